/*
 * orc_admm.c -- CPU ORACLE (test infrastructure, not the product).
 *
 * The reference's ADMM driver restated in plain C:
 *   osqp_set_default_settings / osqp_setup / osqp_solve / osqp_cleanup and the
 *   data updates                         src/osqp.c:24-1332
 *   iteration steps, residuals, tolerances, infeasibility tests, rho logic,
 *   termination, solution storage        src/auxil.c:13-786
 *   box projection / normal cone         src/proj.c:4-29
 *   polish                               src/polish.c:19-350
 *
 * Deliberate, documented deviations (all behaviour-preserving for parity):
 *   - settings->adaptive_rho_interval == 0 is resolved with the reference's
 *     non-PROFILING rule (osqp.c:267-279: 4*check_termination, or 100 when
 *     termination checks are off) instead of the wall-clock heuristic of the
 *     PROFILING build (osqp.c:453-485), which makes iteration counts depend on
 *     timing.  The interval is resolved per solve and not
 *     written back into the settings.
 *   - no printing, no SIGINT handler.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include "orc_osqp.h"
#include "orc_internal.h"

#define ORC_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define ORC_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define INF_BOUND (OSQP_INFTY * MIN_SCALING)   /* 1e26: "infinite" bound test */

double orc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- status bookkeeping (auxil.c:632-679) -------------------------------- */
static void set_status(OSQPInfo *info, c_int v) {
  const char *s = NULL;
  info->status_val = v;
  switch (v) {
  case OSQP_SOLVED:                       s = "solved"; break;
  case OSQP_SOLVED_INACCURATE:            s = "solved inaccurate"; break;
  case OSQP_PRIMAL_INFEASIBLE:            s = "primal infeasible"; break;
  case OSQP_PRIMAL_INFEASIBLE_INACCURATE: s = "primal infeasible inaccurate"; break;
  case OSQP_UNSOLVED:                     s = "unsolved"; break;
  case OSQP_DUAL_INFEASIBLE:              s = "dual infeasible"; break;
  case OSQP_DUAL_INFEASIBLE_INACCURATE:   s = "dual infeasible inaccurate"; break;
  case OSQP_MAX_ITER_REACHED:             s = "maximum iterations reached"; break;
  case OSQP_TIME_LIMIT_REACHED:           s = "run time limit reached"; break;
  case OSQP_SIGINT:                       s = "interrupted"; break;
  case OSQP_NON_CVX:                      s = "problem non convex"; break;
  default: break;
  }
  if (s) { strncpy(info->status, s, sizeof(info->status) - 1); info->status[31] = 0; }
}

static void reset_info(OSQPInfo *info) {
  info->solve_time = 0.0;
  info->polish_time = 0.0;
  set_status(info, OSQP_UNSOLVED);
  info->rho_updates = 0;
}

static c_int has_solution(const OSQPInfo *info) {
  c_int v = info->status_val;
  return v != OSQP_PRIMAL_INFEASIBLE && v != OSQP_PRIMAL_INFEASIBLE_INACCURATE &&
         v != OSQP_DUAL_INFEASIBLE && v != OSQP_DUAL_INFEASIBLE_INACCURATE &&
         v != OSQP_NON_CVX;
}

/* ---- defaults (osqp.c:24-71) --------------------------------------------- */
void orc_osqp_set_default_settings(OSQPSettings *s) {
  s->rho = RHO; s->sigma = SIGMA; s->scaling = SCALING;
  s->adaptive_rho = ADAPTIVE_RHO; s->adaptive_rho_interval = ADAPTIVE_RHO_INTERVAL;
  s->adaptive_rho_tolerance = ADAPTIVE_RHO_TOLERANCE;
  s->adaptive_rho_fraction = ADAPTIVE_RHO_FRACTION;
  s->max_iter = MAX_ITER; s->eps_abs = EPS_ABS; s->eps_rel = EPS_REL;
  s->eps_prim_inf = EPS_PRIM_INF; s->eps_dual_inf = EPS_DUAL_INF;
  s->alpha = ALPHA; s->linsys_solver = QDLDL_SOLVER;
  s->delta = DELTA; s->polish = POLISH; s->polish_refine_iter = POLISH_REFINE_ITER;
  s->verbose = 0;  /* the oracle never prints */
  s->scaled_termination = SCALED_TERMINATION;
  s->check_termination = CHECK_TERMINATION; s->warm_start = WARM_START;
  s->time_limit = TIME_LIMIT;
}

/* ---- validation (auxil.c:791-1065) --------------------------------------- */
static c_int bad_data(const OSQPData *d) {
  if (!d || !d->P || !d->A || !d->q) return 1;
  if (d->n <= 0 || d->m < 0) return 1;
  if (d->P->m != d->n || d->P->m != d->P->n) return 1;
  for (c_int j = 0; j < d->n; j++)
    for (c_int k = d->P->p[j]; k < d->P->p[j + 1]; k++)
      if (d->P->i[k] > j) return 1;
  if (d->A->m != d->m || d->A->n != d->n) return 1;
  for (c_int i = 0; i < d->m; i++) if (d->l[i] > d->u[i]) return 1;
  return 0;
}

static c_int bad_settings(const OSQPSettings *s) {
  if (!s) return 1;
  if (s->scaling < 0) return 1;
  if (s->adaptive_rho != 0 && s->adaptive_rho != 1) return 1;
  if (s->adaptive_rho_interval < 0) return 1;
  if (s->adaptive_rho_fraction <= 0) return 1;
  if (s->adaptive_rho_tolerance < 1.0) return 1;
  if (s->polish_refine_iter < 0) return 1;
  if (s->rho <= 0.0 || s->sigma <= 0.0 || s->delta <= 0.0) return 1;
  if (s->max_iter <= 0) return 1;
  if (s->eps_abs < 0.0 || s->eps_rel < 0.0) return 1;
  if (s->eps_rel == 0.0 && s->eps_abs == 0.0) return 1;
  if (s->eps_prim_inf <= 0.0 || s->eps_dual_inf <= 0.0) return 1;
  if (s->alpha <= 0.0 || s->alpha >= 2.0) return 1;
  if (s->linsys_solver != QDLDL_SOLVER && s->linsys_solver != MKL_PARDISO_SOLVER) return 1;
  if (s->verbose != 0 && s->verbose != 1) return 1;
  if (s->scaled_termination != 0 && s->scaled_termination != 1) return 1;
  if (s->check_termination < 0) return 1;
  if (s->warm_start != 0 && s->warm_start != 1) return 1;
  if (s->time_limit < 0.0) return 1;
  return 0;
}

/* ---- rho vector by constraint class (auxil.c:76-142) --------------------- */
static c_int classify(c_float l, c_float u) {
  if (l < -INF_BOUND && u > INF_BOUND) return -1;  /* loose    */
  if (u - l < RHO_TOL) return 1;                    /* equality */
  return 0;                                         /* inequality */
}

static void set_rho_vec(OSQPWorkspace *w) {
  w->settings->rho = ORC_MIN(ORC_MAX(w->settings->rho, RHO_MIN), RHO_MAX);
  for (c_int i = 0; i < w->data->m; i++) {
    c_int t = classify(w->data->l[i], w->data->u[i]);
    w->constr_type[i] = t;
    w->rho_vec[i] = (t == -1) ? RHO_MIN
                  : (t == 1) ? RHO_EQ_OVER_RHO_INEQ * w->settings->rho : w->settings->rho;
    w->rho_inv_vec[i] = 1. / w->rho_vec[i];
  }
}

static c_int refresh_rho_vec(OSQPWorkspace *w) {
  c_int changed = 0;
  for (c_int i = 0; i < w->data->m; i++) {
    c_int t = classify(w->data->l[i], w->data->u[i]);
    if (t == w->constr_type[i]) continue;
    w->constr_type[i] = t;
    w->rho_vec[i] = (t == -1) ? RHO_MIN
                  : (t == 1) ? RHO_EQ_OVER_RHO_INEQ * w->settings->rho : w->settings->rho;
    w->rho_inv_vec[i] = 1. / w->rho_vec[i];
    changed = 1;
  }
  if (changed) return w->linsys_solver->update_rho_vec(w->linsys_solver, w->rho_vec);
  return 0;
}

void orc_cold_start(OSQPWorkspace *w) {
  orc_vec_fill(w->x, 0., w->data->n);
  orc_vec_fill(w->z, 0., w->data->m);
  orc_vec_fill(w->y, 0., w->data->m);
}

/* ---- setup (osqp.c:76-283) ----------------------------------------------- */
static c_float *zeros(c_int n) { return (c_float *)calloc((size_t)(n > 0 ? n : 1), sizeof(c_float)); }

c_int orc_osqp_setup(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings) {
  if (bad_data(data)) return OSQP_DATA_VALIDATION_ERROR;
  if (bad_settings(settings)) return OSQP_SETTINGS_VALIDATION_ERROR;
  OSQPWorkspace *w = (OSQPWorkspace *)calloc(1, sizeof(OSQPWorkspace));
  if (!w) return OSQP_MEM_ALLOC_ERROR;
  *workp = w;
  c_int n = data->n, m = data->m;
  w->timer = (OSQPTimer *)malloc(sizeof(OSQPTimer));
  w->timer->t0 = orc_now();

  w->data = (OSQPData *)calloc(1, sizeof(OSQPData));
  w->data->n = n; w->data->m = m;
  w->data->P = orc_csc_copy(data->P);
  w->data->A = orc_csc_copy(data->A);
  w->data->q = orc_vec_dup(data->q, n);
  w->data->l = orc_vec_dup(data->l, m);
  w->data->u = orc_vec_dup(data->u, m);

  w->rho_vec = zeros(m); w->rho_inv_vec = zeros(m);
  w->constr_type = (c_int *)calloc((size_t)(m > 0 ? m : 1), sizeof(c_int));
  w->x = zeros(n); w->z = zeros(m); w->xz_tilde = zeros(n + m);
  w->x_prev = zeros(n); w->z_prev = zeros(m); w->y = zeros(m);
  w->Ax = zeros(m); w->Px = zeros(n); w->Aty = zeros(n);
  w->delta_y = zeros(m); w->Atdelta_y = zeros(n);
  w->delta_x = zeros(n); w->Pdelta_x = zeros(n); w->Adelta_x = zeros(m);

  w->settings = (OSQPSettings *)malloc(sizeof(OSQPSettings));
  *w->settings = *settings;

  if (settings->scaling) {
    w->scaling = (OSQPScaling *)calloc(1, sizeof(OSQPScaling));
    w->scaling->D = zeros(n); w->scaling->Dinv = zeros(n);
    w->scaling->E = zeros(m); w->scaling->Einv = zeros(m);
    w->D_temp = zeros(n); w->D_temp_A = zeros(n); w->E_temp = zeros(m);
    orc_scale_data(w);
  }
  set_rho_vec(w);

  c_int rc = orc_init_linsys_solver(&w->linsys_solver, w->data->P, w->data->A,
                                    w->settings->sigma, w->rho_vec, 0);
  if (rc) return rc;

  w->pol = (OSQPPolish *)calloc(1, sizeof(OSQPPolish));
  size_t mm = (size_t)(m > 0 ? m : 1);
  w->pol->Alow_to_A = (c_int *)malloc(mm * sizeof(c_int));
  w->pol->Aupp_to_A = (c_int *)malloc(mm * sizeof(c_int));
  w->pol->A_to_Alow = (c_int *)malloc(mm * sizeof(c_int));
  w->pol->A_to_Aupp = (c_int *)malloc(mm * sizeof(c_int));
  w->pol->x = zeros(n); w->pol->z = zeros(m); w->pol->y = zeros(m);

  w->solution = (OSQPSolution *)calloc(1, sizeof(OSQPSolution));
  w->solution->x = zeros(n); w->solution->y = zeros(m);
  w->info = (OSQPInfo *)calloc(1, sizeof(OSQPInfo));
  set_status(w->info, OSQP_UNSOLVED);
  w->info->setup_time = orc_now() - w->timer->t0;
  w->first_run = 1;
  w->info->rho_estimate = w->settings->rho;
  return 0;
}

/* ---- cleanup (osqp.c:659-757) -------------------------------------------- */
c_int orc_osqp_cleanup(OSQPWorkspace *w) {
  if (!w) return 0;
  if (w->data) {
    orc_csc_free(w->data->P); orc_csc_free(w->data->A);
    free(w->data->q); free(w->data->l); free(w->data->u); free(w->data);
  }
  if (w->scaling) {
    free(w->scaling->D); free(w->scaling->Dinv); free(w->scaling->E); free(w->scaling->Einv);
    free(w->scaling);
  }
  free(w->D_temp); free(w->D_temp_A); free(w->E_temp);
  if (w->linsys_solver && w->linsys_solver->free) w->linsys_solver->free(w->linsys_solver);
  if (w->pol) {
    free(w->pol->Alow_to_A); free(w->pol->Aupp_to_A); free(w->pol->A_to_Alow);
    free(w->pol->A_to_Aupp); free(w->pol->x); free(w->pol->z); free(w->pol->y);
    free(w->pol);
  }
  free(w->rho_vec); free(w->rho_inv_vec); free(w->constr_type);
  free(w->x); free(w->z); free(w->xz_tilde); free(w->x_prev); free(w->z_prev); free(w->y);
  free(w->Ax); free(w->Px); free(w->Aty); free(w->delta_y); free(w->Atdelta_y);
  free(w->delta_x); free(w->Pdelta_x); free(w->Adelta_x);
  free(w->settings);
  if (w->solution) { free(w->solution->x); free(w->solution->y); free(w->solution); }
  free(w->info); free(w->timer); free(w);
  return 0;
}

/* ---- one ADMM iteration (osqp.c:356-370; auxil.c:147-225; proj.c:4-14) ---- */
void orc_admm_iterate(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m;
  c_float sigma = w->settings->sigma, alpha = w->settings->alpha;
  c_float *t;
  t = w->x; w->x = w->x_prev; w->x_prev = t;     /* x_prev <- x (pointer swap) */
  t = w->z; w->z = w->z_prev; w->z_prev = t;

  /* right-hand side and KKT solve (auxil.c:161-183) */
  for (c_int j = 0; j < n; j++) w->xz_tilde[j] = sigma * w->x_prev[j] - w->data->q[j];
  for (c_int i = 0; i < m; i++) w->xz_tilde[n + i] = w->z_prev[i] - w->rho_inv_vec[i] * w->y[i];
  w->linsys_solver->solve(w->linsys_solver, w->xz_tilde);

  /* x and delta_x (auxil.c:185-198) */
  for (c_int j = 0; j < n; j++)
    w->x[j] = alpha * w->xz_tilde[j] + ((c_float)1.0 - alpha) * w->x_prev[j];
  for (c_int j = 0; j < n; j++) w->delta_x[j] = w->x[j] - w->x_prev[j];

  /* z with projection on [l,u] (auxil.c:200-212, proj.c:4-14) */
  for (c_int i = 0; i < m; i++) {
    c_float v = alpha * w->xz_tilde[n + i] + ((c_float)1.0 - alpha) * w->z_prev[i] +
                w->rho_inv_vec[i] * w->y[i];
    v = ORC_MAX(v, w->data->l[i]);
    w->z[i] = ORC_MIN(v, w->data->u[i]);
  }
  /* y and delta_y (auxil.c:214-225) */
  for (c_int i = 0; i < m; i++) {
    w->delta_y[i] = w->rho_vec[i] * (alpha * w->xz_tilde[n + i] +
                                     ((c_float)1.0 - alpha) * w->z_prev[i] - w->z[i]);
    w->y[i] += w->delta_y[i];
  }
}

/* ---- residuals, objective (auxil.c:227-318) ------------------------------ */
static c_int unscaled_norms(const OSQPWorkspace *w) {
  return w->settings->scaling && !w->settings->scaled_termination;
}

static c_float objective(OSQPWorkspace *w, const c_float *x) {
  c_float v = orc_quad_form(w->data->P, x) + orc_vec_dot(w->data->q, x, w->data->n);
  if (w->settings->scaling) v *= w->scaling->cinv;
  return v;
}

static c_float primal_residual(OSQPWorkspace *w, const c_float *x, const c_float *z) {
  c_int m = w->data->m;
  orc_mat_vec(w->data->A, x, w->Ax, 0);
  for (c_int i = 0; i < m; i++) w->z_prev[i] = w->Ax[i] + (-1.0) * z[i];  /* z_prev is scratch */
  if (unscaled_norms(w)) return orc_vec_scaled_norm_inf(w->scaling->Einv, w->z_prev, m);
  return orc_vec_norm_inf(w->z_prev, m);
}

static c_float dual_residual(OSQPWorkspace *w, const c_float *x, const c_float *y) {
  c_int n = w->data->n;
  memcpy(w->x_prev, w->data->q, (size_t)n * sizeof(c_float));            /* x_prev is scratch */
  orc_mat_vec(w->data->P, x, w->Px, 0);
  orc_mat_tpose_vec(w->data->P, x, w->Px, 1, 1);
  for (c_int j = 0; j < n; j++) w->x_prev[j] = w->x_prev[j] + w->Px[j];
  if (w->data->m > 0) {
    orc_mat_tpose_vec(w->data->A, y, w->Aty, 0, 0);
    for (c_int j = 0; j < n; j++) w->x_prev[j] = w->x_prev[j] + w->Aty[j];
  }
  if (unscaled_norms(w))
    return w->scaling->cinv * orc_vec_scaled_norm_inf(w->scaling->Dinv, w->x_prev, n);
  return orc_vec_norm_inf(w->x_prev, n);
}

/* auxil.c:564-629 */
void orc_update_info(OSQPWorkspace *w, c_int iter, c_int compute_objective, c_int polish) {
  c_float *x, *y, *z, *obj, *pri, *dua;
  if (polish) {
    x = w->pol->x; y = w->pol->y; z = w->pol->z;
    obj = &w->pol->obj_val; pri = &w->pol->pri_res; dua = &w->pol->dua_res;
  } else {
    x = w->x; y = w->y; z = w->z;
    obj = &w->info->obj_val; pri = &w->info->pri_res; dua = &w->info->dua_res;
    w->info->iter = iter;
  }
  if (compute_objective) *obj = objective(w, x);
  *pri = (w->data->m == 0) ? 0. : primal_residual(w, x, z);
  *dua = dual_residual(w, x, y);
  if (polish) w->info->polish_time = orc_now() - w->timer->t0;
  else        w->info->solve_time  = orc_now() - w->timer->t0;
}

/* ---- tolerances (auxil.c:256-285, 320-359) ------------------------------- */
static c_float primal_tol(const OSQPWorkspace *w, c_float eps_abs, c_float eps_rel) {
  c_int m = w->data->m;
  c_float a, b;
  if (unscaled_norms(w)) {
    a = orc_vec_scaled_norm_inf(w->scaling->Einv, w->z, m);
    b = orc_vec_scaled_norm_inf(w->scaling->Einv, w->Ax, m);
  } else {
    a = orc_vec_norm_inf(w->z, m);
    b = orc_vec_norm_inf(w->Ax, m);
  }
  return eps_abs + eps_rel * ORC_MAX(a, b);
}

static c_float dual_tol(const OSQPWorkspace *w, c_float eps_abs, c_float eps_rel) {
  c_int n = w->data->n;
  c_float v, t;
  if (unscaled_norms(w)) {
    v = orc_vec_scaled_norm_inf(w->scaling->Dinv, w->data->q, n);
    t = orc_vec_scaled_norm_inf(w->scaling->Dinv, w->Aty, n); v = ORC_MAX(v, t);
    t = orc_vec_scaled_norm_inf(w->scaling->Dinv, w->Px, n);  v = ORC_MAX(v, t);
    v *= w->scaling->cinv;
  } else {
    v = orc_vec_norm_inf(w->data->q, n);
    t = orc_vec_norm_inf(w->Aty, n); v = ORC_MAX(v, t);
    t = orc_vec_norm_inf(w->Px, n);  v = ORC_MAX(v, t);
  }
  return eps_abs + eps_rel * v;
}

/* ---- infeasibility certificates (auxil.c:361-512) ------------------------ */
static c_int primal_infeasible(OSQPWorkspace *w, c_float eps) {
  c_int n = w->data->n, m = w->data->m;
  const c_float *l = w->data->l, *u = w->data->u;
  c_float *dy = w->delta_y;
  for (c_int i = 0; i < m; i++) {               /* project on the recession cone's polar */
    if (u[i] > INF_BOUND) {
      if (l[i] < -INF_BOUND) dy[i] = 0.0;
      else dy[i] = ORC_MIN(dy[i], 0.0);
    } else if (l[i] < -INF_BOUND) {
      dy[i] = ORC_MAX(dy[i], 0.0);
    }
  }
  c_float nrm;
  if (unscaled_norms(w)) {
    orc_vec_ew_prod(w->scaling->E, dy, w->Adelta_x, m);   /* Adelta_x is scratch */
    nrm = orc_vec_norm_inf(w->Adelta_x, m);
  } else nrm = orc_vec_norm_inf(dy, m);
  if (nrm > OSQP_DIVISION_TOL) {
    c_float lhs = 0.0;
    for (c_int i = 0; i < m; i++) lhs += u[i] * ORC_MAX(dy[i], 0) + l[i] * ORC_MIN(dy[i], 0);
    if (lhs < eps * nrm) {
      orc_mat_tpose_vec(w->data->A, dy, w->Atdelta_y, 0, 0);
      if (unscaled_norms(w)) orc_vec_ew_prod(w->scaling->Dinv, w->Atdelta_y, w->Atdelta_y, n);
      return orc_vec_norm_inf(w->Atdelta_y, n) < eps * nrm;
    }
  }
  return 0;
}

static c_int dual_infeasible(OSQPWorkspace *w, c_float eps) {
  c_int n = w->data->n, m = w->data->m;
  c_float nrm, cs;
  if (unscaled_norms(w)) {
    nrm = orc_vec_scaled_norm_inf(w->scaling->D, w->delta_x, n);
    cs = w->scaling->c;
  } else { nrm = orc_vec_norm_inf(w->delta_x, n); cs = 1.0; }
  if (!(nrm > OSQP_DIVISION_TOL)) return 0;
  if (!(orc_vec_dot(w->data->q, w->delta_x, n) < cs * eps * nrm)) return 0;
  orc_mat_vec(w->data->P, w->delta_x, w->Pdelta_x, 0);
  orc_mat_tpose_vec(w->data->P, w->delta_x, w->Pdelta_x, 1, 1);
  if (unscaled_norms(w)) orc_vec_ew_prod(w->scaling->Dinv, w->Pdelta_x, w->Pdelta_x, n);
  if (!(orc_vec_norm_inf(w->Pdelta_x, n) < cs * eps * nrm)) return 0;
  orc_mat_vec(w->data->A, w->delta_x, w->Adelta_x, 0);
  if (unscaled_norms(w)) orc_vec_ew_prod(w->scaling->Einv, w->Adelta_x, w->Adelta_x, m);
  for (c_int i = 0; i < m; i++) {
    if ((w->data->u[i] < INF_BOUND && w->Adelta_x[i] > eps * nrm) ||
        (w->data->l[i] > -INF_BOUND && w->Adelta_x[i] < -eps * nrm)) return 0;
  }
  return 1;
}

/* ---- termination (auxil.c:681-786) --------------------------------------- */
static c_int check_termination(OSQPWorkspace *w, c_int approximate) {
  c_float eps_abs = w->settings->eps_abs, eps_rel = w->settings->eps_rel;
  c_float eps_pinf = w->settings->eps_prim_inf, eps_dinf = w->settings->eps_dual_inf;
  c_int prim_ok = 0, dual_ok = 0, pinf = 0, dinf = 0;

  if (w->info->pri_res > OSQP_INFTY || w->info->dua_res > OSQP_INFTY) {
    set_status(w->info, OSQP_NON_CVX);
    w->info->obj_val = OSQP_NAN;
    return 1;
  }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; eps_pinf *= 10; eps_dinf *= 10; }

  if (w->data->m == 0) prim_ok = 1;
  else if (w->info->pri_res < primal_tol(w, eps_abs, eps_rel)) prim_ok = 1;
  else pinf = primal_infeasible(w, eps_pinf);

  if (w->info->dua_res < dual_tol(w, eps_abs, eps_rel)) dual_ok = 1;
  else dinf = dual_infeasible(w, eps_dinf);

  if (prim_ok && dual_ok) {
    set_status(w->info, approximate ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED);
    return 1;
  }
  if (pinf) {
    set_status(w->info, approximate ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE);
    if (unscaled_norms(w)) orc_vec_ew_prod(w->scaling->E, w->delta_y, w->delta_y, w->data->m);
    w->info->obj_val = OSQP_INFTY;
    return 1;
  }
  if (dinf) {
    set_status(w->info, approximate ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE);
    if (unscaled_norms(w)) orc_vec_ew_prod(w->scaling->D, w->delta_x, w->delta_x, w->data->n);
    w->info->obj_val = -OSQP_INFTY;
    return 1;
  }
  return 0;
}

/* ---- rho adaptation (auxil.c:13-74) -------------------------------------- */
static c_float rho_estimate(const OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m;
  /* z_prev / x_prev hold the scaled residual vectors left by update_info */
  c_float pri = orc_vec_norm_inf(w->z_prev, m);
  c_float dua = orc_vec_norm_inf(w->x_prev, n);
  c_float pn = ORC_MAX(orc_vec_norm_inf(w->z, m), orc_vec_norm_inf(w->Ax, m));
  pri /= (pn + OSQP_DIVISION_TOL);
  c_float dn = ORC_MAX(orc_vec_norm_inf(w->data->q, n), orc_vec_norm_inf(w->Aty, n));
  dn = ORC_MAX(dn, orc_vec_norm_inf(w->Px, n));
  dua /= (dn + OSQP_DIVISION_TOL);
  c_float r = w->settings->rho * sqrt(pri / dua);
  return ORC_MIN(ORC_MAX(r, RHO_MIN), RHO_MAX);
}

static c_int adapt_rho(OSQPWorkspace *w) {
  c_float r = rho_estimate(w);
  c_int rc = 0;
  w->info->rho_estimate = r;
  if (r > w->settings->rho * w->settings->adaptive_rho_tolerance ||
      r < w->settings->rho / w->settings->adaptive_rho_tolerance) {
    rc = orc_osqp_update_rho(w, r);
    w->info->rho_updates += 1;
  }
  return rc;
}

/* ---- solution extraction (auxil.c:524-562; scaling.c:177-192) ------------ */
static void store_solution(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m;
  if (has_solution(w->info)) {
    memcpy(w->solution->x, w->x, (size_t)n * sizeof(c_float));
    memcpy(w->solution->y, w->y, (size_t)m * sizeof(c_float));
    if (w->settings->scaling) {
      orc_vec_ew_prod(w->scaling->D, w->solution->x, w->solution->x, n);
      orc_vec_ew_prod(w->scaling->E, w->solution->y, w->solution->y, m);
      orc_vec_scale(w->solution->y, w->scaling->cinv, m);
    }
  } else {
    orc_vec_fill(w->solution->x, OSQP_NAN, n);
    orc_vec_fill(w->solution->y, OSQP_NAN, m);
    c_int v = w->info->status_val;
    if (v == OSQP_PRIMAL_INFEASIBLE || v == OSQP_PRIMAL_INFEASIBLE_INACCURATE)
      orc_vec_scale(w->delta_y, 1. / orc_vec_norm_inf(w->delta_y, m), m);
    if (v == OSQP_DUAL_INFEASIBLE || v == OSQP_DUAL_INFEASIBLE_INACCURATE)
      orc_vec_scale(w->delta_x, 1. / orc_vec_norm_inf(w->delta_x, n), n);
    orc_cold_start(w);
  }
}

/* ---- solve (osqp.c:288-654) ----------------------------------------------- */
c_int orc_osqp_solve(OSQPWorkspace *w) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (w->clear_update_time == 1) w->info->update_time = 0.0;
  w->rho_update_from_solve = 1;
  c_int can_check = 0, iter, exitflag = 0;
  const OSQPSettings *st = w->settings;
  w->timer->t0 = orc_now();

  /* deterministic stand-in for the timing-driven automatic interval */
  c_int rho_interval = st->adaptive_rho_interval;
  if (st->adaptive_rho && !rho_interval)
    rho_interval = st->check_termination ? ADAPTIVE_RHO_MULTIPLE_TERMINATION * st->check_termination
                                         : ADAPTIVE_RHO_FIXED;

  if (!st->warm_start) orc_cold_start(w);

  for (iter = 1; iter <= st->max_iter; iter++) {
    orc_admm_iterate(w);

    if (st->time_limit) {
      c_float t = (w->first_run ? w->info->setup_time : w->info->update_time) +
                  (orc_now() - w->timer->t0);
      if (t >= st->time_limit) { set_status(w->info, OSQP_TIME_LIMIT_REACHED); break; }
    }
    can_check = st->check_termination && (iter % st->check_termination == 0);
    if (can_check) {
      orc_update_info(w, iter, 0, 0);
      if (check_termination(w, 0)) break;
    }
    if (st->adaptive_rho && rho_interval && (iter % rho_interval == 0)) {
      if (!can_check) orc_update_info(w, iter, 0, 0);
      if (adapt_rho(w)) { exitflag = 1; goto out; }
    }
  }

  if (!can_check) {
    orc_update_info(w, iter - 1, 0, 0);
    check_termination(w, 0);
  }
  if (has_solution(w->info)) w->info->obj_val = objective(w, w->x);

  if (w->info->status_val == OSQP_UNSOLVED) {
    if (!check_termination(w, 1)) set_status(w->info, OSQP_MAX_ITER_REACHED);
  }
  if (w->info->status_val == OSQP_TIME_LIMIT_REACHED) {
    if (!check_termination(w, 1)) set_status(w->info, OSQP_TIME_LIMIT_REACHED);
  }
  w->info->rho_estimate = rho_estimate(w);
  w->info->solve_time = orc_now() - w->timer->t0;

  if (st->polish && w->info->status_val == OSQP_SOLVED) orc_polish(w);

  w->info->run_time = (w->first_run ? w->info->setup_time : w->info->update_time) +
                      w->info->solve_time + w->info->polish_time;
  w->first_run = 0;
  w->clear_update_time = 1;
  w->rho_update_from_solve = 0;
  store_solution(w);
out:
  return exitflag;
}

/* ---- data updates (osqp.c:765-1007) -------------------------------------- */
static void update_timer_begin(OSQPWorkspace *w) {
  if (w->clear_update_time == 1) { w->clear_update_time = 0; w->info->update_time = 0.0; }
  w->timer->t0 = orc_now();
}
static void update_timer_end(OSQPWorkspace *w) { w->info->update_time += orc_now() - w->timer->t0; }

c_int orc_osqp_update_lin_cost(OSQPWorkspace *w, const c_float *q_new) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  update_timer_begin(w);
  c_int n = w->data->n;
  memcpy(w->data->q, q_new, (size_t)n * sizeof(c_float));
  if (w->settings->scaling) {
    orc_vec_ew_prod(w->scaling->D, w->data->q, w->data->q, n);
    orc_vec_scale(w->data->q, w->scaling->c, n);
  }
  reset_info(w->info);
  update_timer_end(w);
  return 0;
}

c_int orc_osqp_update_bounds(OSQPWorkspace *w, const c_float *l_new, const c_float *u_new) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  update_timer_begin(w);
  c_int m = w->data->m;
  for (c_int i = 0; i < m; i++) if (l_new[i] > u_new[i]) return 1;
  memcpy(w->data->l, l_new, (size_t)m * sizeof(c_float));
  memcpy(w->data->u, u_new, (size_t)m * sizeof(c_float));
  if (w->settings->scaling) {
    orc_vec_ew_prod(w->scaling->E, w->data->l, w->data->l, m);
    orc_vec_ew_prod(w->scaling->E, w->data->u, w->data->u, m);
  }
  reset_info(w->info);
  c_int rc = refresh_rho_vec(w);
  update_timer_end(w);
  return rc;
}

c_int orc_osqp_update_lower_bound(OSQPWorkspace *w, const c_float *l_new) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  update_timer_begin(w);
  c_int m = w->data->m;
  memcpy(w->data->l, l_new, (size_t)m * sizeof(c_float));
  if (w->settings->scaling) orc_vec_ew_prod(w->scaling->E, w->data->l, w->data->l, m);
  for (c_int i = 0; i < m; i++) if (w->data->l[i] > w->data->u[i]) return 1;
  reset_info(w->info);
  c_int rc = refresh_rho_vec(w);
  update_timer_end(w);
  return rc;
}

c_int orc_osqp_update_upper_bound(OSQPWorkspace *w, const c_float *u_new) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  update_timer_begin(w);
  c_int m = w->data->m;
  memcpy(w->data->u, u_new, (size_t)m * sizeof(c_float));
  if (w->settings->scaling) orc_vec_ew_prod(w->scaling->E, w->data->u, w->data->u, m);
  for (c_int i = 0; i < m; i++) if (w->data->u[i] < w->data->l[i]) return 1;
  reset_info(w->info);
  c_int rc = refresh_rho_vec(w);
  update_timer_end(w);
  return rc;
}

c_int orc_osqp_warm_start(OSQPWorkspace *w, const c_float *x, const c_float *y) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  c_int rc = orc_osqp_warm_start_y(w, y);
  if (rc) return rc;
  return orc_osqp_warm_start_x(w, x);
}

c_int orc_osqp_warm_start_x(OSQPWorkspace *w, const c_float *x) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!w->settings->warm_start) w->settings->warm_start = 1;
  c_int n = w->data->n;
  memcpy(w->x, x, (size_t)n * sizeof(c_float));
  if (w->settings->scaling) orc_vec_ew_prod(w->scaling->Dinv, w->x, w->x, n);
  orc_mat_vec(w->data->A, w->x, w->z, 0);
  return 0;
}

c_int orc_osqp_warm_start_y(OSQPWorkspace *w, const c_float *y) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (!w->settings->warm_start) w->settings->warm_start = 1;
  c_int m = w->data->m;
  memcpy(w->y, y, (size_t)m * sizeof(c_float));
  if (w->settings->scaling) {
    orc_vec_ew_prod(w->scaling->Einv, w->y, w->y, m);
    orc_vec_scale(w->y, w->scaling->c, m);
  }
  return 0;
}

/* ---- matrix updates (osqp.c:1012-1279) ----------------------------------- */
static c_int patch_matrices(OSQPWorkspace *w, const c_float *Px_new, const c_int *Px_idx,
                            c_int P_n, int do_P, const c_float *Ax_new,
                            const c_int *Ax_idx, c_int A_n, int do_A) {
  c_int nnzP = w->data->P->p[w->data->P->n], nnzA = w->data->A->p[w->data->A->n];
  update_timer_begin(w);
  if (do_P && Px_idx && P_n > nnzP) return 1;
  if (do_A && Ax_idx && A_n > nnzA) return do_P ? 2 : 1;
  if (w->settings->scaling) orc_unscale_data(w);
  if (do_P) {
    if (Px_idx) for (c_int k = 0; k < P_n; k++) w->data->P->x[Px_idx[k]] = Px_new[k];
    else        for (c_int k = 0; k < nnzP; k++) w->data->P->x[k] = Px_new[k];
  }
  if (do_A) {
    if (Ax_idx) for (c_int k = 0; k < A_n; k++) w->data->A->x[Ax_idx[k]] = Ax_new[k];
    else        for (c_int k = 0; k < nnzA; k++) w->data->A->x[k] = Ax_new[k];
  }
  if (w->settings->scaling) orc_scale_data(w);
  c_int rc = w->linsys_solver->update_matrices(w->linsys_solver, w->data->P, w->data->A);
  reset_info(w->info);
  update_timer_end(w);
  return rc;
}

c_int orc_osqp_update_P(OSQPWorkspace *w, const c_float *Px_new, const c_int *Px_idx, c_int P_n) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  return patch_matrices(w, Px_new, Px_idx, P_n, 1, NULL, NULL, 0, 0);
}
c_int orc_osqp_update_A(OSQPWorkspace *w, const c_float *Ax_new, const c_int *Ax_idx, c_int A_n) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  return patch_matrices(w, NULL, NULL, 0, 0, Ax_new, Ax_idx, A_n, 1);
}
c_int orc_osqp_update_P_A(OSQPWorkspace *w, const c_float *Px_new, const c_int *Px_idx, c_int P_n,
                          const c_float *Ax_new, const c_int *Ax_idx, c_int A_n) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  return patch_matrices(w, Px_new, Px_idx, P_n, 1, Ax_new, Ax_idx, A_n, 1);
}

/* osqp.c:1281-1332 */
c_int orc_osqp_update_rho(OSQPWorkspace *w, c_float rho_new) {
  if (!w) return OSQP_WORKSPACE_NOT_INIT_ERROR;
  if (rho_new <= 0) return 1;
  double t0 = 0;
  if (!w->rho_update_from_solve) {
    if (w->clear_update_time == 1) { w->clear_update_time = 0; w->info->update_time = 0.0; }
    t0 = orc_now();
  }
  w->settings->rho = ORC_MIN(ORC_MAX(rho_new, RHO_MIN), RHO_MAX);
  for (c_int i = 0; i < w->data->m; i++) {
    if (w->constr_type[i] == 0) {
      w->rho_vec[i] = w->settings->rho;
      w->rho_inv_vec[i] = 1. / w->settings->rho;
    } else if (w->constr_type[i] == 1) {
      w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->settings->rho;
      w->rho_inv_vec[i] = 1. / w->rho_vec[i];
    }
  }
  c_int rc = w->linsys_solver->update_rho_vec(w->linsys_solver, w->rho_vec);
  if (!w->rho_update_from_solve) w->info->update_time += orc_now() - t0;
  return rc;
}

/* ---- polish (polish.c:19-350) -------------------------------------------- */
static c_int build_Ared(OSQPWorkspace *w) {
  OSQPPolish *p = w->pol;
  c_int m = w->data->m, n = w->data->n;
  const csc *A = w->data->A;
  p->n_low = p->n_upp = 0;
  for (c_int i = 0; i < m; i++) {
    if (w->z[i] - w->data->l[i] < -w->y[i]) { p->Alow_to_A[p->n_low] = i; p->A_to_Alow[i] = p->n_low++; }
    else p->A_to_Alow[i] = -1;
  }
  for (c_int i = 0; i < m; i++) {
    if (w->data->u[i] - w->z[i] < w->y[i]) { p->Aupp_to_A[p->n_upp] = i; p->A_to_Aupp[i] = p->n_upp++; }
    else p->A_to_Aupp[i] = -1;
  }
  c_int mred = p->n_low + p->n_upp, cnt = 0;
  for (c_int k = 0; k < A->p[n]; k++)
    if (p->A_to_Alow[A->i[k]] != -1 || p->A_to_Aupp[A->i[k]] != -1) cnt++;
  p->Ared = orc_csc_alloc(mred, n, cnt, 1, 0);
  if (!p->Ared) return -1;
  cnt = 0;
  for (c_int j = 0; j < n; j++) {
    p->Ared->p[j] = cnt;
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      c_int r = A->i[k];
      if (p->A_to_Alow[r] != -1)      { p->Ared->i[cnt] = p->A_to_Alow[r];            p->Ared->x[cnt++] = A->x[k]; }
      else if (p->A_to_Aupp[r] != -1) { p->Ared->i[cnt] = p->A_to_Aupp[r] + p->n_low; p->Ared->x[cnt++] = A->x[k]; }
    }
  }
  p->Ared->p[n] = cnt;
  return mred;
}

c_int orc_polish(OSQPWorkspace *w) {
  OSQPPolish *p = w->pol;
  c_int n = w->data->n, m = w->data->m;
  w->timer->t0 = orc_now();
  c_int mred = build_Ared(w);
  if (mred < 0) { w->info->status_polish = -1; return -1; }
  LinSysSolver *ls = NULL;
  if (orc_init_linsys_solver(&ls, w->data->P, p->Ared, w->settings->delta, NULL, 1)) {
    w->info->status_polish = -1;
    orc_csc_free(p->Ared);
    return 1;
  }
  c_int N = n + mred;
  c_float *rhs = zeros(N), *sol = zeros(N), *res = zeros(N);
  for (c_int j = 0; j < n; j++) rhs[j] = -w->data->q[j];
  for (c_int k = 0; k < p->n_low; k++) rhs[n + k] = w->data->l[p->Alow_to_A[k]];
  for (c_int k = 0; k < p->n_upp; k++) rhs[n + p->n_low + k] = w->data->u[p->Aupp_to_A[k]];
  memcpy(sol, rhs, (size_t)N * sizeof(c_float));
  ls->solve(ls, sol);
  /* iterative refinement against the unregularised KKT (polish.c:134-181) */
  for (c_int it = 0; it < w->settings->polish_refine_iter; it++) {
    memcpy(res, rhs, (size_t)N * sizeof(c_float));
    orc_mat_vec(w->data->P, sol, res, -1);
    orc_mat_tpose_vec(w->data->P, sol, res, -1, 1);
    orc_mat_tpose_vec(p->Ared, sol + n, res, -1, 0);
    orc_mat_vec(p->Ared, sol, res + n, -1);
    ls->solve(ls, res);
    for (c_int k = 0; k < N; k++) sol[k] += res[k];
  }
  memcpy(p->x, sol, (size_t)n * sizeof(c_float));
  orc_mat_vec(w->data->A, p->x, p->z, 0);
  for (c_int i = 0; i < m; i++) {
    if (mred == 0) p->y[i] = 0.0;
    else if (p->A_to_Alow[i] != -1) p->y[i] = sol[n + p->A_to_Alow[i]];
    else if (p->A_to_Aupp[i] != -1) p->y[i] = sol[n + p->n_low + p->A_to_Aupp[i]];
    else p->y[i] = 0.0;
  }
  /* (z,y) onto the normal cone (proj.c:16-29); z_prev is scratch */
  for (c_int i = 0; i < m; i++) {
    w->z_prev[i] = p->z[i] + p->y[i];
    p->z[i] = ORC_MIN(ORC_MAX(w->z_prev[i], w->data->l[i]), w->data->u[i]);
    p->y[i] = w->z_prev[i] - p->z[i];
  }
  orc_update_info(w, 0, 1, 1);
  c_int ok = (p->pri_res < w->info->pri_res && p->dua_res < w->info->dua_res) ||
             (p->pri_res < w->info->pri_res && w->info->dua_res < 1e-10) ||
             (p->dua_res < w->info->dua_res && w->info->pri_res < 1e-10);
  if (ok) {
    w->info->obj_val = p->obj_val; w->info->pri_res = p->pri_res; w->info->dua_res = p->dua_res;
    w->info->status_polish = 1;
    memcpy(w->x, p->x, (size_t)n * sizeof(c_float));
    memcpy(w->z, p->z, (size_t)m * sizeof(c_float));
    memcpy(w->y, p->y, (size_t)m * sizeof(c_float));
  } else w->info->status_polish = -1;
  ls->free(ls);
  orc_csc_free(p->Ared);
  free(rhs); free(sol); free(res);
  return 0;
}
