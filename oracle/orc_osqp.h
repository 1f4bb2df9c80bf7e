/*
 * orc_osqp.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's CPU path for osqp_setup / osqp_solve /
 * osqp_update_* and its direct LDL^T linear-system plugin.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (osqp_amd/csrc) never links or calls it.
 *
 * Every function cites the reference file:line it follows.  All entry points
 * carry an orc_ prefix so the oracle can live in one process beside the
 * product library (which exports the reference's own names).
 *
 * Parity pinning (see DESIGN.md "Oracle"): the reference cannot be built in
 * this image (its QDLDL submodule is empty and osqp_configure.h is cmake
 * generated), so the oracle is pinned by the reference's own fixtures captured
 * in tests/golden (JSON files) (expected solutions, KKT known-answer vectors,
 * lin_alg vectors) and by the known answers recorded in SURVEY.md App. B-D.
 * Bit-level parity with QDLDL's factor is unpinned (any correct LDL^T passes
 * the reference's own test_solveKKT).
 */
#ifndef ORC_OSQP_H
#define ORC_OSQP_H

#include "../include/osqp_amd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sparse / dense kernels (src/lin_alg.c, src/cs.c) ------------------- */
void    orc_mat_vec(const csc *A, const c_float *x, c_float *y, c_int plus_eq);
void    orc_mat_tpose_vec(const csc *A, const c_float *x, c_float *y,
                          c_int plus_eq, c_int skip_diag);
c_float orc_quad_form(const csc *P, const c_float *x);
c_float orc_vec_norm_inf(const c_float *v, c_int l);
c_float orc_vec_scaled_norm_inf(const c_float *S, const c_float *v, c_int l);
void    orc_mat_inf_norm_cols(const csc *M, c_float *E);
void    orc_mat_inf_norm_rows(const csc *M, c_float *E);
void    orc_mat_inf_norm_cols_sym_triu(const csc *M, c_float *E);
csc    *orc_csc_alloc(c_int m, c_int n, c_int nzmax, c_int values, c_int triplet);
void    orc_csc_free(csc *A);
csc    *orc_csc_copy(const csc *A);
csc    *orc_csc_view(c_int m, c_int n, c_int nzmax, c_float *x, c_int *i, c_int *p);

/* ---- KKT assembly (src/kkt.c) ------------------------------------------- */
csc *orc_form_KKT(const csc *P, const csc *A, c_float param1, const c_float *param2,
                  c_int *PtoKKT, c_int *AtoKKT, c_int *param2toKKT);

/* ---- LDL^T kernel with the QDLDL v0.1.5 call contract -------------------- */
c_int orc_ldl_etree(c_int n, const c_int *Ap, const c_int *Ai, c_int *work,
                    c_int *Lnz, c_int *etree);
c_int orc_ldl_factor(c_int n, const c_int *Ap, const c_int *Ai, const c_float *Ax,
                     c_int *Lp, c_int *Li, c_float *Lx, c_float *D, c_float *Dinv,
                     const c_int *Lnz, const c_int *etree, unsigned char *bwork,
                     c_int *iwork, c_float *fwork);
void  orc_ldl_solve(c_int n, const c_int *Lp, const c_int *Li, const c_float *Lx,
                    const c_float *Dinv, c_float *x);
/* fill-reducing ordering of a symmetric matrix given by its upper triangle */
c_int orc_min_degree_order(c_int n, const c_int *Ap, const c_int *Ai, c_int *perm);

/* ---- direct plugin (lin_sys/direct/qdldl/qdldl_interface.c) -------------- */
c_int orc_init_linsys_solver(LinSysSolver **s, const csc *P, const csc *A,
                             c_float sigma, const c_float *rho_vec, c_int polish);
c_int orc_linsys_nnzL(const LinSysSolver *s);
/* bench.py: per-phase times of the direct path's setup with a time budget for the numeric factorisation (orc_ldl.c) */
c_int orc_ldl_phase_times(const csc *P, const csc *A, c_float sigma, const c_float *rho_vec, c_float budget_s, c_float out[8]);

/* ---- public API (src/osqp.c) -------------------------------------------- */
void  orc_osqp_set_default_settings(OSQPSettings *settings);
c_int orc_osqp_setup(OSQPWorkspace **workp, const OSQPData *data,
                     const OSQPSettings *settings);
c_int orc_osqp_solve(OSQPWorkspace *work);
c_int orc_osqp_cleanup(OSQPWorkspace *work);
c_int orc_osqp_update_lin_cost(OSQPWorkspace *work, const c_float *q_new);
c_int orc_osqp_update_bounds(OSQPWorkspace *work, const c_float *l_new,
                             const c_float *u_new);
c_int orc_osqp_update_lower_bound(OSQPWorkspace *work, const c_float *l_new);
c_int orc_osqp_update_upper_bound(OSQPWorkspace *work, const c_float *u_new);
c_int orc_osqp_warm_start(OSQPWorkspace *work, const c_float *x, const c_float *y);
c_int orc_osqp_warm_start_x(OSQPWorkspace *work, const c_float *x);
c_int orc_osqp_warm_start_y(OSQPWorkspace *work, const c_float *y);
c_int orc_osqp_update_P(OSQPWorkspace *work, const c_float *Px_new,
                        const c_int *Px_new_idx, c_int P_new_n);
c_int orc_osqp_update_A(OSQPWorkspace *work, const c_float *Ax_new,
                        const c_int *Ax_new_idx, c_int A_new_n);
c_int orc_osqp_update_P_A(OSQPWorkspace *work, const c_float *Px_new,
                          const c_int *Px_new_idx, c_int P_new_n,
                          const c_float *Ax_new, const c_int *Ax_new_idx,
                          c_int A_new_n);
c_int orc_osqp_update_rho(OSQPWorkspace *work, c_float rho_new);
void  orc_cold_start(OSQPWorkspace *work);

/* One ADMM iteration on the oracle workspace (osqp.c:356-370); exposed so the
 * per-iteration trajectory of the HIP engine can be compared step by step. */
void  orc_admm_iterate(OSQPWorkspace *work);
/* update_info without timing (auxil.c:564-629) */
void  orc_update_info(OSQPWorkspace *work, c_int iter, c_int compute_objective,
                      c_int polish);

#ifdef __cplusplus
}
#endif
#endif
