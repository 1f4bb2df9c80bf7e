/*
 * osqp_amd.h -- public C API of libosqp_amd.so, the MI355X-native drop-in for
 * the reference's osqp_setup / osqp_solve / osqp_update_* path.
 *
 * Same names, argument meaning, return codes and struct layouts as the
 * reference's include/osqp.h (cited per function, paths relative to
 * /root/reference).  The ADMM loop, the KKT solve (indirect: Jacobi-PCG on the
 * reduced system) and every residual reduction run on the GPU; there is no CPU
 * fallback -- without a HIP device osqp_setup fails with
 * OSQP_LINSYS_SOLVER_LOAD_ERROR and says why on stderr.
 */
#ifndef OSQP_AMD_H
#define OSQP_AMD_H

#include "osqp_amd_types.h"
#include "osqp_amd_helpers.h"   /* cs.h / lin_alg.h / kkt.h helper symbols, allocator hook */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- main API (include/osqp.h:32-96; src/osqp.c:24-757) ------------------ */
void  osqp_set_default_settings(OSQPSettings *settings);
c_int osqp_setup(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings);
c_int osqp_solve(OSQPWorkspace *work);
c_int osqp_cleanup(OSQPWorkspace *work);

/* ---- data updates (include/osqp.h:108-258; src/osqp.c:765-1332) ---------- */
c_int osqp_update_lin_cost(OSQPWorkspace *work, const c_float *q_new);
c_int osqp_update_bounds(OSQPWorkspace *work, const c_float *l_new, const c_float *u_new);
c_int osqp_update_lower_bound(OSQPWorkspace *work, const c_float *l_new);
c_int osqp_update_upper_bound(OSQPWorkspace *work, const c_float *u_new);
c_int osqp_warm_start(OSQPWorkspace *work, const c_float *x, const c_float *y);
c_int osqp_warm_start_x(OSQPWorkspace *work, const c_float *x);
c_int osqp_warm_start_y(OSQPWorkspace *work, const c_float *y);
c_int osqp_update_P(OSQPWorkspace *work, const c_float *Px_new, const c_int *Px_new_idx, c_int P_new_n);
c_int osqp_update_A(OSQPWorkspace *work, const c_float *Ax_new, const c_int *Ax_new_idx, c_int A_new_n);
c_int osqp_update_P_A(OSQPWorkspace *work, const c_float *Px_new, const c_int *Px_new_idx, c_int P_new_n,
                      const c_float *Ax_new, const c_int *Ax_new_idx, c_int A_new_n);
c_int osqp_update_rho(OSQPWorkspace *work, c_float rho_new);

/* ---- settings setters (include/osqp.h:270-421; src/osqp.c:1339-1617) ----- */
c_int osqp_update_max_iter(OSQPWorkspace *work, c_int max_iter_new);
c_int osqp_update_eps_abs(OSQPWorkspace *work, c_float eps_abs_new);
c_int osqp_update_eps_rel(OSQPWorkspace *work, c_float eps_rel_new);
c_int osqp_update_eps_prim_inf(OSQPWorkspace *work, c_float eps_prim_inf_new);
c_int osqp_update_eps_dual_inf(OSQPWorkspace *work, c_float eps_dual_inf_new);
c_int osqp_update_alpha(OSQPWorkspace *work, c_float alpha_new);
c_int osqp_update_warm_start(OSQPWorkspace *work, c_int warm_start_new);
c_int osqp_update_scaled_termination(OSQPWorkspace *work, c_int scaled_termination_new);
c_int osqp_update_check_termination(OSQPWorkspace *work, c_int check_termination_new);
c_int osqp_update_delta(OSQPWorkspace *work, c_float delta_new);
c_int osqp_update_polish(OSQPWorkspace *work, c_int polish_new);
c_int osqp_update_polish_refine_iter(OSQPWorkspace *work, c_int polish_refine_iter_new);
c_int osqp_update_verbose(OSQPWorkspace *work, c_int verbose_new);
c_int osqp_update_time_limit(OSQPWorkspace *work, c_float time_limit_new);

/* ---- helpers callers of the reference use (include/auxil.h; the cs.h / lin_alg.h / kkt.h
 *      ones are declared in osqp_amd_helpers.h) ------------------------------------------ */
void  cold_start(OSQPWorkspace *work);                                           /* auxil.h:62 */

/* ---- linear-system plugin boundary (include/lin_sys.h:20-45) -------------
 * Constructor of the HIP PCG plugin, in the shape of
 * init_linsys_solver_qdldl (lin_sys/direct/qdldl/qdldl_interface.h:89).  The
 * returned object starts with the reference's vtable prefix
 * (include/types.h:298-319). */
c_int init_linsys_solver_hip_pcg(LinSysSolver **sp, const csc *P, const csc *A,
                                 c_float sigma, const c_float *rho_vec, c_int polish);
c_int load_linsys_solver(enum linsys_solver_type linsys_solver);      /* lin_sys.c:15 */
c_int unload_linsys_solver(enum linsys_solver_type linsys_solver);    /* lin_sys.c:35 */
c_int init_linsys_solver(LinSysSolver **s, const csc *P, const csc *A, c_float sigma,
                         const c_float *rho_vec, enum linsys_solver_type linsys_solver,
                         c_int polish);                                /* lin_sys.c:56 */

/* ---- side channel for knobs OSQPSettings has no room for ------------------
 * (OSQPSettings is ABI; SURVEY.md section 5 "config / flags").  Process-wide
 * defaults picked up by the next osqp_setup / init_linsys_solver_hip_pcg;
 * also settable through the environment:
 *   OSQP_AMD_PCG_EPS_REL (default 1e-9), OSQP_AMD_PCG_EPS_ABS (1e-15),
 *   OSQP_AMD_PCG_MAX_ITER (0 = max(20000, 10n): see DESIGN.md), OSQP_AMD_DEVICE (0). */
typedef struct {
  c_float pcg_eps_rel;
  c_float pcg_eps_abs;
  c_int   pcg_max_iter;
  c_int   device;
  c_int   pcg_adaptive;   /* 1: inexact solves, PCG tolerance tied to the ADMM residuals (not parity-exact; OSQP_AMD_PCG_ADAPTIVE) */
} osqp_amd_options;
void  osqp_amd_get_options(osqp_amd_options *opt);
void  osqp_amd_set_options(const osqp_amd_options *opt);
/* Statistics of the device engine behind a workspace (PCG iteration counts,
 * graph launches); returns nonzero if the workspace has no engine. */
typedef struct {
  c_int pcg_iters_total;
  c_int pcg_iters_last;
  c_int pcg_forced;
  c_int graph_launches;
  c_int host_syncs;
  c_int resident;        /* 1: the linear solves run as one resident launch each (K in registers, engine.hip k_pcg_resident);
                            0: launch-per-step PCG kernels (problem too large for the register files, OSQP_AMD_RESIDENT=0,
                            or a resident launch found the GPU shared and the engine fell back) */
} osqp_amd_stats;
c_int osqp_amd_get_stats(const OSQPWorkspace *work, osqp_amd_stats *st);
/* The options above are the DEFAULTS a new workspace / plugin instance copies at creation; afterwards
 * each instance has its own (no knob is shared between live workspaces).  These two read / change the
 * copy of one workspace (the device cannot be changed after setup). */
c_int osqp_amd_get_workspace_options(const OSQPWorkspace *work, osqp_amd_options *opt);
c_int osqp_amd_set_workspace_options(OSQPWorkspace *work, const osqp_amd_options *opt);
/* Binary problem files: one little-endian file per QP (layout in osqp_host.c); the data the
 * reference's generators emit as C headers (tests/utils/codegen_utils.py:172-347).
 * Return 0, or 1 bad argument / 2 cannot open / 3 malformed / 4 out of memory. */
c_int osqp_amd_write_problem(const char *path, const OSQPData *data);
c_int osqp_amd_read_problem(const char *path, OSQPData **data);
void  osqp_amd_free_problem(OSQPData *data);
/* Raw engine handle of a workspace (for the kernel-level tests / bench). */
void *osqp_amd_engine(const OSQPWorkspace *work);

#ifdef __cplusplus
}
#endif
#endif
