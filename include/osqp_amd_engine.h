/*
 * osqp_amd_engine.h -- thin C-ABI shim between the C host code and the HIP
 * (gfx950) device engine.  Plain pointers and sizes only; no C++ or torch
 * types.  Each entry point names the reference function(s) it replaces
 * (paths relative to /root/reference).
 *
 * One `hipeng` = one QP resident on one GPU, with its own HIP stream; no
 * process-global state (unlike lin_sys/direct/pardiso/pardiso_loader.c:29-32).
 * All functions return 0 on success or a negative HIPENG_* code; they never
 * fall back to the CPU.
 */
#ifndef OSQP_AMD_ENGINE_H
#define OSQP_AMD_ENGINE_H

#include "osqp_amd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HIPENG_OK            0
#define HIPENG_ERR_NO_DEVICE (-101)  /* no HIP device / runtime failure at create */
#define HIPENG_ERR_HIP       (-102)  /* a HIP call failed (message on stderr)      */
#define HIPENG_ERR_ARG       (-103)
#define HIPENG_ERR_ALLOC     (-104)

typedef struct hipeng hipeng;

/* Mutable scalar parameters read by the kernels at launch. */
typedef struct {
  c_float sigma;        /* settings->sigma                                  */
  c_float alpha;        /* settings->alpha                                  */
  c_float pcg_eps_rel;  /* PCG stops at ||r||_2 <= max(eps_rel*||b||_2, eps_abs) */
  c_float pcg_eps_abs;
  c_int   pcg_max_iter; /* hard cap per linear solve                        */
  c_int   no_restart;   /* 1: any loss of positivity in the CG recurrence counts as negative
                           curvature (setup-time convexity probe); 0: restart at the rounding floor */
} hipeng_params;

/* Scalars produced by one residual evaluation (replaces the reductions inside
 * update_info / compute_pri_res / compute_dua_res / compute_pri_tol /
 * compute_dua_tol / compute_rho_estimate, src/auxil.c:13-52, 227-359, and the
 * cheap parts of is_primal_infeasible / is_dual_infeasible, :361-512). */
typedef struct {
  c_float pri_res_u, pri_res_s;     /* ||Einv(Ax-z)||, ||Ax-z||              */
  c_float z_u, z_s, Ax_u, Ax_s;     /* ||Einv z||, ||z||, ||Einv Ax||, ||Ax|| */
  c_float dua_res_u, dua_res_s;     /* ||Dinv(q+Px+A'y)|| (no cinv), scaled  */
  c_float q_u, q_s, Aty_u, Aty_s, Px_u, Px_s;
  c_float obj_scaled;               /* 1/2 x'Px + q'x in scaled space         */
  c_float dy_norm_u, dy_norm_s;     /* ||E dy_proj||, ||dy_proj||             */
  c_float dy_lhs;                   /* u'max(dy,0) + l'min(dy,0)              */
  c_float dx_norm_u, dx_norm_s;     /* ||D dx||, ||dx||                       */
  c_float q_dx;                     /* q'dx                                   */
  /* filled by hipeng_certificates only */
  c_float Atdy_u, Atdy_s;           /* ||Dinv A'dy_proj||, ||A'dy_proj||       */
  c_float Pdx_u, Pdx_s;             /* ||Dinv P dx||, ||P dx||                 */
  c_float Adx_viol;                 /* number of rows violating the A dx test */
} hipeng_scalars;

typedef struct {
  c_int admm_done;       /* iterations completed by the last hipeng_run_admm  */
  c_int pcg_iters_total; /* PCG iterations since create/reset                 */
  c_int pcg_iters_last;  /* PCG iterations of the most recent linear solve    */
  c_int pcg_iters_max;   /* max per solve since the last hipeng_run_admm call */
  c_int pcg_forced;      /* solves accepted at pcg_max_iter without converging */
  c_int graph_launches;
  c_int kernels_per_pcg_iter;
  c_int neg_curvature;   /* solves in which CG met p'Kp <= 0 (K not positive definite) */
  c_int resident;        /* 1: linear solves as resident launches (k_pcg_resident) */
} hipeng_stats;

/* Create the device-resident problem.  P (upper triangle) and A are the
 * workspace's scaled matrices in CSC with int64 indices; they are re-indexed
 * to int32 and uploaded as CSR(A) and the fused row matrix [P | A'].
 * Replaces the allocation + init part of init_linsys_solver
 * (src/lin_sys.c:56-75, lin_sys/direct/qdldl/qdldl_interface.c:177-323). */
int hipeng_create(hipeng **out, const csc *P, const csc *A, const c_float *q,
                  const c_float *l, const c_float *u, const c_float *rho_vec,
                  const hipeng_params *prm, int device);
void hipeng_destroy(hipeng *e);

int hipeng_set_params(hipeng *e, const hipeng_params *prm);
/* D, E, Dinv, Einv vectors (NULL = identity) and cost scaling c: needed only
 * for the unscaled norms of hipeng_residuals (src/scaling.c:177-192). */
int hipeng_set_scaling(hipeng *e, const c_float *D, const c_float *E, c_float c);
/* scale_data (src/scaling.c:44-156) on the device: Ruiz-equilibrates the resident raw problem
 * in place (`passes` = settings->scaling) and returns D, E, c and the scaled q, l, u, triu(P) and A
 * values (CSC order of the matrices given to hipeng_create) for the host mirrors. */
int hipeng_ruiz_scale(hipeng *e, c_int passes, c_float *D, c_float *E, c_float *cost,
                      c_float *q, c_float *l, c_float *u, c_float *Px, c_float *Ax);
int hipeng_matrices_changed(hipeng *e);   /* after an in-place rescale: preconditioner, z~, rhs parts */
int hipeng_upload_q(hipeng *e, const c_float *q);                      /* osqp_update_lin_cost, osqp.c:765 */
int hipeng_upload_bounds(hipeng *e, const c_float *l, const c_float *u); /* osqp_update_bounds, osqp.c:797 */
/* update_rho_vec of the vtable (qdldl_interface.c:396-410): re-uploads rho and
 * rebuilds the Jacobi preconditioner diag(P)+sigma+sum_i rho_i A_ij^2. */
int hipeng_upload_rho(hipeng *e, const c_float *rho_vec);
/* update_matrices of the vtable (qdldl_interface.c:381-393): new values, same
 * sparsity. */
int hipeng_upload_matrices(hipeng *e, const csc *P, const csc *A);
/* cold_start (auxil.c:155) / osqp_warm_start* (osqp.c:942-1007): x, y NULL =
 * keep; z is recomputed as A x on the device when x is given. */
int hipeng_cold_start(hipeng *e);
int hipeng_set_iterates(hipeng *e, const c_float *x, const c_float *y);
/* overwrite z (and the PCG's z~ warm-start image) -- used when polish adopts
 * its (x, z, y) (src/polish.c:321-325) */
int hipeng_set_z(hipeng *e, const c_float *z);

/* Run `count` ADMM iterations (osqp.c:356-370: swap, update_xz_tilde,
 * update_x, update_z, update_y) entirely on the device. */
int hipeng_run_admm(hipeng *e, c_int count);
/* Evaluate all residual scalars for the current iterates (device reductions,
 * one small D2H copy). */
int hipeng_residuals(hipeng *e, hipeng_scalars *out);
/* Second stage of the infeasibility tests: A'dy, P dx, A dx (auxil.c:401-417,
 * 456-497).  eps_dx = eps_dual_inf * ||dx|| threshold for the A dx row test;
 * unscaled != 0 applies Einv to A dx first. */
int hipeng_certificates(hipeng *e, c_float eps_dx, int unscaled, hipeng_scalars *inout);
/* Copy device vectors to host arrays (NULL = skip).  dy is the projected
 * delta_y when `dy_projected` is nonzero. */
int hipeng_download(hipeng *e, c_float *x, c_float *y, c_float *z, c_float *dx,
                    c_float *dy, int dy_projected);
int hipeng_get_stats(hipeng *e, hipeng_stats *st);
int hipeng_reset_stats(hipeng *e);   /* zero the counters (after the setup-time convexity probe) */
int hipeng_sync(hipeng *e);

/* Plugin-boundary solve (LinSysSolver.solve, include/types.h:300-301;
 * qdldl_interface.c:350-376): b = [sigma x - q ; z - y/rho] on the host is
 * overwritten by [x_tilde ; z_tilde]. */
int hipeng_kkt_solve(hipeng *e, c_float *b);

/* Kernel-level entry points used by the parity tests (host in, host out):
 * which = 0: y = A x   (mat_vec, lin_alg.c:241-271)
 *         1: y = A' x  (mat_tpose_vec, lin_alg.c:273-322)
 *         2: y = P x   (mat_vec + mat_tpose_vec skip_diag on triu P) */
int hipeng_spmv(hipeng *e, int which, const c_float *x, c_float *y);
/* The same products on device pointers (d_x, d_y: n or m doubles in HBM), enqueued on the engine's stream
 * without a sync -- hipeng_sync() orders them before other streams read d_y.  Building block of the
 * row-partitioned multi-GPU variant. */
int hipeng_spmv_dev(hipeng *e, int which, const c_float *d_x, c_float *d_y);
/* Time `reps` back-to-back launches of one hot kernel on the engine stream with
 * HIP events; returns average microseconds per launch in *usec.
 * which = 0: k_cg_A, 1: k_cg_B, 3 / 4: k_cg_A update-only / apply-only (split mode), 5: k_pcg_init,
 * 6: k_cg_A + k_cg_B in loop order, 7: an empty dependent launch, 8: k_pcg_init + k_pcg_resident (the
 * right-hand side and the whole linear solve of one ADMM iteration; resident engines only). */
int hipeng_time_kernel(hipeng *e, int which, int reps, double *usec);
/* Algorithmic bytes of one launch of the kernels above. */
int hipeng_kernel_bytes(hipeng *e, int which, double *bytes);
int hipeng_is_split(hipeng *e);   /* k_cg_A as two launches (update-only + apply-only)? */
/* Slack-like variables (one entry in their column of A, no coupling in P) are eliminated from the linear system exactly
 * (engine.hip, k_elim_refresh; launch-per-step engines).  hipeng_row_eliminated: does row i of A carry such a variable?
 * hipeng_elim_count: how many variables are eliminated. */
int hipeng_row_eliminated(hipeng *e, c_int i);
c_int hipeng_elim_count(hipeng *e);
/* Resident PCG (the reduced matrix K = P + sigma I + A' rho A held in registers, one launch per linear
 * solve; engine.hip, k_pcg_resident).  out[0] structures built, [1] in use, [2] entries of K per thread,
 * [3] workgroups, [4] nnz(K), [5] LDS bytes per workgroup, [6] PCG iterations of the most recent linear
 * solve, [7] pipelined recurrences switched off for the current K, [8] true-residual checks that failed since create
 * (each continued its solve from the true residual), [9] form (1: k_pcg_resident, 2: k_pcg_blockres), [10] launches that
 * gave up waiting since create (each sends the rest of its run_admm call to the launch-per-step kernels), [11] waits inside
 * resident launches that ended well but took more than 50 us, [12] the longest of them in ticks of the 100 MHz clock,
 * [13] flags / granules stored again by a workgroup whose own wait went on, [14] give-ups in a row (the third ends the mode
 * for this engine; a run_admm call without one starts the count again), [15] block-direct form: coupling rows of A carried as the low-rank term (0: the plain form, huge rows only).  [1] stays 1 until that third one. */
int hipeng_resident_info(hipeng *e, long long out[16]);
/* For the CPU tests (needs no device): the host side of the resident set-up -- symbolic K, row partition, positions in
 * the exchanged vector, register layout -- for a grid of exactly `nwg` workgroups (nwg > 0) or, nwg < 0, with the grid sized
 * to the problem as the engine does on a machine of -nwg CUs (stats[4] then tells the grid).  stats: [0] qualifies, [1] entries of K per thread,
 * [2] length of the exchanged vector, [3] nnz(K), [4] workgroups that own rows, [5] most rows, [6] most entries per
 * workgroup, [7] threads that hold entries (448).  Optional outputs may be NULL. */
int hipeng_resident_plan(const csc *P, const csc *A, int nwg, long long stats[8], int *Kptr, int *Kcol, int *kdst, long long cap,
                         unsigned short *rowpos, unsigned short *slotcol, int *wg4);
/* For the tests: K as the resident kernel holds it, as triplets; returns nnz(K) or a negative code. */
long long hipeng_resident_dump(hipeng *e, int *row, int *col, double *val, long long cap);

/* Dense-direct solve (engine.hip res_kind 4, csrc/dense_direct.h): test / measurement hook for its blocked inversion.
 * A: n x n row-major symmetric positive definite host matrix, n a multiple of 128; Ainv: its inverse (explicit, by blocked
 * Gauss-Jordan on v_mfma_f64_16x16x4_f64); ms[0]: device time of the inversion, ms[1]: of one n x n x n GEMM of the same
 * kernel.  Returns 0, 1 when a pivot was not positive, or a HIPENG_ERR_* code. */
int hipeng_dense_invert_selftest(int n, const double *A, double *Ainv, double *ms);

#ifdef __cplusplus
}
#endif
#endif
