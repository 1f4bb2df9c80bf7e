/* osqp_amd_rowpart.h -- ONE QP solved over several GPUs by rows (BASELINE config 5
 * "1 -> 8 MI355X"), driven from C.
 *
 * The reference has no counterpart: its linear solve is one thread (lin_sys/direct/qdldl/qdldl_interface.c:216) and a
 * GPU / indirect solver is a TODO (ROADMAP.md:1-3).  What this replaces is therefore the reference's osqp_solve loop
 * itself (src/osqp.c:354-532: compute_rhs / solve / update_x / update_z / update_y, src/auxil.c:161-225; residuals and
 * termination src/auxil.c:240-359, 681-740; rho adaptation src/auxil.c:13-74) for a problem whose rows of A and
 * columns of triu(P) are sharded over `world` ranks, one process per GPU:
 *
 *   - every rank holds the n-vectors (x, x~, q, the PCG vectors) replicated and the m-vectors (z, y, l, u, rho) of its
 *     own rows only; its shard (P_g, A_g) is resident in an ordinary engine (hipeng, set up with scaling = 0 on
 *     already scaled data);
 *   - the data-path traffic is ONE all-reduce of an n-vector per PCG iteration (K u = sigma u + sum_g [P_g u +
 *     A_g' rho_g (A_g u)]), one per ADMM iteration for the right-hand side, and two n-vectors + six scalars at a
 *     termination check.  Every rank holds the same bits after an all-reduce, so the PCG scalars are computed
 *     redundantly on each device and need no collective of their own;
 *   - the whole loop -- kernels, collectives, the decision when a PCG solve has converged -- is issued from
 *     osqp_amd_rp_solve on the shard engine's stream; the host reads one flag per group of PCG iterations and a
 *     handful of scalars per termination check.
 *
 * The collective is a callback, so that the caller's process group decides how ranks talk: the built-in provider
 * (osqp_amd_rp_use_rccl) calls ncclAllReduce of librccl.so on the engine's stream (RCCL over xGMI); a test harness can
 * pass any function with the same meaning (tests: torch.distributed with gloo).
 *
 * Statuses: OSQP_SOLVED, OSQP_SOLVED_INACCURATE, OSQP_MAX_ITER_REACHED (no infeasibility certificates and no polish in
 * this variant).  All functions return 0 or a HIPENG_ERR_* code. */
#ifndef OSQP_AMD_ROWPART_H
#define OSQP_AMD_ROWPART_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct osqp_amd_rp osqp_amd_rp;

/* In-place all-reduce of `count` doubles at device address `buf` over all ranks; op 0 = sum, 1 = max.  Must be ordered
 * after the work already in `hip_stream` and before work submitted to it later (a stream-ordered collective on that
 * stream, or a synchronous one that synchronises the stream itself).  Returns 0 on success. */
typedef int (*osqp_amd_rp_allreduce_fn)(void *user, void *buf, long long count, int op, void *hip_stream);

typedef struct {
  double rho, sigma, alpha, eps_abs, eps_rel, adaptive_rho_tolerance, pcg_eps_rel;
  int max_iter, check_termination, adaptive_rho, adaptive_rho_interval, scaled_termination, pcg_max_iter;
} osqp_amd_rp_settings;                 /* meaning and defaults of the fields: OSQPSettings (include/osqp_amd_types.h) */

typedef struct {
  int status;                           /* OSQP_SOLVED = 1, OSQP_SOLVED_INACCURATE = 2, OSQP_MAX_ITER_REACHED = -2 */
  int iter, rho_updates;
  long long pcg_iters, collectives;
  double obj_val, pri_res, dua_res, rho_estimate;
} osqp_amd_rp_info;

/* shard_engine: a hipeng (osqp_amd_engine.h) that holds this rank's (P_g, A_g) with scaling = 0: n columns, m_loc rows.
 * q, D: n doubles; l_loc, u_loc, E_loc: m_loc doubles (the scaled problem and its scaling, src/scaling.c:44-156);
 * c: cost scaling; has_eq_any: some rank has an equality row (tightens the PCG stop as osqp_solve does). */
osqp_amd_rp *osqp_amd_rp_create(void *shard_engine, const double *q, const double *l_loc, const double *u_loc,
                                const double *D, const double *E_loc, double c, int m_total, int has_eq_any,
                                const osqp_amd_rp_settings *settings, int world, int rank,
                                osqp_amd_rp_allreduce_fn allreduce, void *user);
/* Replace the callback by ncclAllReduce on the engine's stream.  unique_id: the ncclUniqueId bytes of rank 0
 * (osqp_amd_rp_rccl_unique_id), the same on every rank.  Collective: every rank calls it.  (NULL, 0) detaches again. */
int  osqp_amd_rp_rccl_unique_id(void *out, int cap_bytes);          /* returns the number of bytes written, < 0 on error */
int  osqp_amd_rp_use_rccl(osqp_amd_rp *rp, const void *unique_id, int id_bytes);
int  osqp_amd_rp_solve(osqp_amd_rp *rp, osqp_amd_rp_info *info);
/* x: n doubles (unscaled, the same on every rank); y_loc: m_loc doubles (unscaled duals of this rank's rows) */
int  osqp_amd_rp_get_solution(osqp_amd_rp *rp, double *x, double *y_loc);
void osqp_amd_rp_free(osqp_amd_rp *rp);

#ifdef __cplusplus
}
#endif
#endif
