/*
 * osqp_amd_batch.h -- C ABI of the batched engine: many independent QPs that
 * share one sparsity pattern (MPC-style, BASELINE config 4), one workgroup per
 * QP on the GPU.  The reference has no batch API: semantically every QP of the
 * batch goes through what osqp_setup + osqp_solve (src/osqp.c:76-654) do for a
 * single problem, with the same settings struct; results are per-QP
 * OSQPInfo-like records.  Plain pointers and sizes only.
 */
#ifndef OSQP_AMD_BATCH_H
#define OSQP_AMD_BATCH_H

#include "osqp_amd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct osqp_amd_batch osqp_amd_batch;

/* P (upper triangle) and A give the shared CSC pattern (and the shared values
 * when Px_all / Ax_all are NULL).  Px_all [batch][nnzP] / Ax_all [batch][nnzA]
 * optionally give per-QP values.  Q [batch][n], L, U [batch][m] row-major.
 * Returns 0 or an osqp_error_type code (include/constants.h:42-50 numbering). */
c_int osqp_amd_batch_setup(osqp_amd_batch **out, c_int batch, const csc *P, const csc *A,
                           const c_float *Px_all, const c_float *Ax_all,
                           const c_float *Q, const c_float *L, const c_float *U,
                           const OSQPSettings *settings, c_int device);
/* osqp_update_lin_cost / osqp_update_bounds for every QP (NULL = keep). */
c_int osqp_amd_batch_update(osqp_amd_batch *b, const c_float *Q, const c_float *L, const c_float *U);
/* osqp_solve for every QP; iterates persist on the device between calls
 * (warm start, settings->warm_start). */
c_int osqp_amd_batch_solve(osqp_amd_batch *b);
/* Results: X [batch][n], Y [batch][m] (unscaled; OSQP_NAN when infeasible),
 * info8 [batch][8] = {iter, status_val, obj_val, pri_res, dua_res, rho_updates,
 * rho_estimate, rho}; DX / DY infeasibility certificates.  NULL = skip. */
c_int osqp_amd_batch_get(osqp_amd_batch *b, c_float *X, c_float *Y, c_float *info8,
                         c_float *DX, c_float *DY);
/* Device pointers of the result arrays, for device-side gathers (RCCL). */
c_int osqp_amd_batch_device_ptrs(osqp_amd_batch *b, void **X, void **Y, void **info8);
void  osqp_amd_batch_cleanup(osqp_amd_batch *b);

#ifdef __cplusplus
}
#endif
#endif
