/*
 * orc_linalg.c -- CPU ORACLE (test infrastructure, not the product).
 * Sparse/dense kernels and CSC helpers restated from the reference:
 *   src/lin_alg.c (vector ops :7-201, matrix scalings :209-239, SpMV :241-322,
 *   row/col norms :325-382, quad_form :387-413) and src/cs.c (alloc/copy/
 *   triplet compression :12-123, symmetric permutation :153-206).
 * Summation orders are kept identical to the reference loops so results can be
 * compared bit for bit (build with -ffp-contract=off).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc_osqp.h"
#include "orc_internal.h"

/* ---- dense vector reductions (lin_alg.c:19-55,143-152) ------------------- */
c_float orc_vec_norm_inf(const c_float *v, c_int l) {
  c_float best = 0.0;
  for (c_int k = 0; k < l; k++) {
    c_float a = v[k] < 0 ? -v[k] : v[k];
    if (a > best) best = a;
  }
  return best;
}

c_float orc_vec_scaled_norm_inf(const c_float *S, const c_float *v, c_int l) {
  c_float best = 0.0;
  for (c_int k = 0; k < l; k++) {
    c_float a = S[k] * v[k];
    if (a < 0) a = -a;
    if (a > best) best = a;
  }
  return best;
}

c_float orc_vec_dot(const c_float *a, const c_float *b, c_int n) {
  c_float acc = 0.0;
  for (c_int k = 0; k < n; k++) acc += a[k] * b[k];
  return acc;
}

c_float orc_vec_mean(const c_float *a, c_int n) {
  c_float acc = 0.0;
  for (c_int k = 0; k < n; k++) acc += a[k];
  return acc / (c_float)n;
}

void orc_vec_fill(c_float *a, c_float v, c_int n) {
  for (c_int k = 0; k < n; k++) a[k] = v;
}

void orc_vec_ew_prod(const c_float *a, const c_float *b, c_float *c, c_int n) {
  for (c_int k = 0; k < n; k++) c[k] = b[k] * a[k];
}

void orc_vec_scale(c_float *a, c_float s, c_int n) {
  for (c_int k = 0; k < n; k++) a[k] *= s;
}

c_float *orc_vec_dup(const c_float *a, c_int n) {
  c_float *b = (c_float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(c_float));
  if (b && n > 0) memcpy(b, a, (size_t)n * sizeof(c_float));
  return b;
}

/* ---- SpMV in CSC (lin_alg.c:241-322) -------------------------------------
 * y (=, +=, -=) A x : column scatter, so each y[i] accumulates its row in
 * increasing column order. */
void orc_mat_vec(const csc *A, const c_float *x, c_float *y, c_int plus_eq) {
  if (!plus_eq) for (c_int i = 0; i < A->m; i++) y[i] = 0;
  if (A->p[A->n] == 0) return;
  if (plus_eq == -1) {
    for (c_int j = 0; j < A->n; j++)
      for (c_int k = A->p[j]; k < A->p[j + 1]; k++) y[A->i[k]] -= A->x[k] * x[j];
  } else {
    for (c_int j = 0; j < A->n; j++)
      for (c_int k = A->p[j]; k < A->p[j + 1]; k++) y[A->i[k]] += A->x[k] * x[j];
  }
}

/* y (=, +=, -=) A' x : column gather; skip_diag drops (j,j) entries so that
 * P x = triu(P) x + triu(P)' x without counting the diagonal twice. */
void orc_mat_tpose_vec(const csc *A, const c_float *x, c_float *y,
                       c_int plus_eq, c_int skip_diag) {
  if (!plus_eq) for (c_int j = 0; j < A->n; j++) y[j] = 0;
  if (A->p[A->n] == 0) return;
  c_float sgn = (plus_eq == -1) ? -1.0 : 1.0;
  for (c_int j = 0; j < A->n; j++) {
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      c_int i = A->i[k];
      c_float t = (skip_diag && i == j) ? 0.0 : A->x[k] * x[i];
      if (sgn > 0) y[j] += t; else y[j] -= t;
    }
  }
}

/* (1/2) x' P x from the upper triangle (lin_alg.c:387-413) */
c_float orc_quad_form(const csc *P, const c_float *x) {
  c_float acc = 0.0;
  for (c_int j = 0; j < P->n; j++) {
    for (c_int k = P->p[j]; k < P->p[j + 1]; k++) {
      c_int i = P->i[k];
      if (i == j)      acc += (c_float).5 * P->x[k] * x[i] * x[i];
      else if (i < j)  acc += P->x[k] * x[i] * x[j];
      else return 0.0; /* not upper triangular */
    }
  }
  return acc;
}

/* ---- row / column infinity norms (lin_alg.c:325-382) --------------------- */
void orc_mat_inf_norm_cols(const csc *M, c_float *E) {
  for (c_int j = 0; j < M->n; j++) {
    c_float best = 0.0;
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++) {
      c_float a = fabs(M->x[k]);
      if (a > best) best = a;
    }
    E[j] = best;
  }
}

void orc_mat_inf_norm_rows(const csc *M, c_float *E) {
  for (c_int i = 0; i < M->m; i++) E[i] = 0.0;
  for (c_int j = 0; j < M->n; j++)
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++) {
      c_float a = fabs(M->x[k]);
      if (a > E[M->i[k]]) E[M->i[k]] = a;
    }
}

void orc_mat_inf_norm_cols_sym_triu(const csc *M, c_float *E) {
  for (c_int j = 0; j < M->n; j++) E[j] = 0.0;
  for (c_int j = 0; j < M->n; j++)
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++) {
      c_int i = M->i[k];
      c_float a = fabs(M->x[k]);
      if (a > E[j]) E[j] = a;
      if (i != j && a > E[i]) E[i] = a;
    }
}

/* ---- diagonal scalings of a CSC matrix (lin_alg.c:209-239) --------------- */
void orc_mat_scale(csc *A, c_float s) {
  c_int nnz = A->p[A->n];
  for (c_int k = 0; k < nnz; k++) A->x[k] *= s;
}

void orc_mat_premult_diag(csc *A, const c_float *d) {
  for (c_int j = 0; j < A->n; j++)
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= d[A->i[k]];
}

void orc_mat_postmult_diag(csc *A, const c_float *d) {
  for (c_int j = 0; j < A->n; j++)
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= d[j];
}

/* ---- CSC containers (cs.c:12-53, 208-236) -------------------------------- */
csc *orc_csc_alloc(c_int m, c_int n, c_int nzmax, c_int values, c_int triplet) {
  csc *A = (csc *)calloc(1, sizeof(csc));
  if (!A) return NULL;
  if (nzmax < 1) nzmax = 1;
  A->m = m; A->n = n; A->nzmax = nzmax; A->nz = triplet ? 0 : -1;
  A->p = (c_int *)calloc((size_t)(triplet ? nzmax : n + 1), sizeof(c_int));
  A->i = (c_int *)calloc((size_t)nzmax, sizeof(c_int));
  A->x = values ? (c_float *)calloc((size_t)nzmax, sizeof(c_float)) : NULL;
  if (!A->p || !A->i || (values && !A->x)) { orc_csc_free(A); return NULL; }
  return A;
}

void orc_csc_free(csc *A) {
  if (!A) return;
  free(A->p); free(A->i); free(A->x); free(A);
}

csc *orc_csc_copy(const csc *A) {
  c_int nnz = A->p[A->n];
  csc *B = orc_csc_alloc(A->m, A->n, nnz, 1, 0);
  if (!B) return NULL;
  memcpy(B->p, A->p, (size_t)(A->n + 1) * sizeof(c_int));
  memcpy(B->i, A->i, (size_t)nnz * sizeof(c_int));
  memcpy(B->x, A->x, (size_t)nnz * sizeof(c_float));
  return B;
}

/* non-owning view over caller arrays (cs.c:12-26, csc_matrix) */
csc *orc_csc_view(c_int m, c_int n, c_int nzmax, c_float *x, c_int *i, c_int *p) {
  csc *A = (csc *)malloc(sizeof(csc));
  if (!A) return NULL;
  A->m = m; A->n = n; A->nz = -1; A->nzmax = nzmax; A->x = x; A->i = i; A->p = p;
  return A;
}

/* triplet -> CSC, stable inside a column (cs.c:55-88); map[k] = CSC slot of
 * triplet entry k */
csc *orc_triplet_to_csc(const csc *T, c_int *map) {
  c_int nz = T->nz, n = T->n;
  csc *C = orc_csc_alloc(T->m, n, nz, 1, 0);
  c_int *next = (c_int *)calloc((size_t)n + 1, sizeof(c_int));
  if (!C || !next) { orc_csc_free(C); free(next); return NULL; }
  for (c_int k = 0; k < nz; k++) C->p[T->p[k] + 1]++;
  for (c_int j = 0; j < n; j++) C->p[j + 1] += C->p[j];
  memcpy(next, C->p, (size_t)n * sizeof(c_int));
  for (c_int k = 0; k < nz; k++) {
    c_int dst = next[T->p[k]]++;
    C->i[dst] = T->i[k];
    C->x[dst] = T->x[k];
    if (map) map[k] = dst;
  }
  free(next);
  return C;
}

/* C = upper(P A P') for symmetric A given by its upper triangle
 * (cs.c:153-206); pinv[old] = new; AtoC[k] = slot of entry k in C. */
csc *orc_symperm_triu(const csc *A, const c_int *pinv, c_int *AtoC) {
  c_int n = A->n, nnz = A->p[n];
  csc *C = orc_csc_alloc(n, n, nnz, 1, 0);
  c_int *cnt = (c_int *)calloc((size_t)n + 1, sizeof(c_int));
  if (!C || !cnt) { orc_csc_free(C); free(cnt); return NULL; }
  for (c_int j = 0; j < n; j++) {
    c_int j2 = pinv[j];
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      c_int i = A->i[k];
      if (i > j) continue;
      c_int i2 = pinv[i];
      cnt[(i2 > j2 ? i2 : j2)]++;
    }
  }
  c_int run = 0;
  for (c_int j = 0; j < n; j++) { C->p[j] = run; run += cnt[j]; cnt[j] = C->p[j]; }
  C->p[n] = run;
  for (c_int j = 0; j < n; j++) {
    c_int j2 = pinv[j];
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      c_int i = A->i[k];
      if (i > j) continue;
      c_int i2 = pinv[i];
      c_int col = i2 > j2 ? i2 : j2, row = i2 < j2 ? i2 : j2;
      c_int q = cnt[col]++;
      C->i[q] = row;
      C->x[q] = A->x[k];
      if (AtoC) AtoC[k] = q;
    }
  }
  free(cnt);
  return C;
}

/* ---- Ruiz equilibration (src/scaling.c:7-156) ---------------------------- */
static void clamp_scaling(c_float *D, c_int n) {
  for (c_int k = 0; k < n; k++) {
    if (D[k] < MIN_SCALING) D[k] = 1.0;
    if (D[k] > MAX_SCALING) D[k] = MAX_SCALING;
  }
}

void orc_scale_data(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m;
  OSQPScaling *s = w->scaling;
  csc *P = w->data->P, *A = w->data->A;
  c_float *q = w->data->q;

  s->c = 1.0;
  orc_vec_fill(s->D, 1., n);    orc_vec_fill(s->Dinv, 1., n);
  orc_vec_fill(s->E, 1., m);    orc_vec_fill(s->Einv, 1., m);

  for (c_int pass = 0; pass < w->settings->scaling; pass++) {
    /* column norms of [P A'; A 0] (scaling.c:28-42) */
    orc_mat_inf_norm_cols_sym_triu(P, w->D_temp);
    orc_mat_inf_norm_cols(A, w->D_temp_A);
    for (c_int j = 0; j < n; j++)
      if (w->D_temp_A[j] > w->D_temp[j]) w->D_temp[j] = w->D_temp_A[j];
    orc_mat_inf_norm_rows(A, w->E_temp);

    clamp_scaling(w->D_temp, n);
    clamp_scaling(w->E_temp, m);
    for (c_int j = 0; j < n; j++) w->D_temp[j] = (c_float)1.0 / sqrt(w->D_temp[j]);
    for (c_int i = 0; i < m; i++) w->E_temp[i] = (c_float)1.0 / sqrt(w->E_temp[i]);

    orc_mat_premult_diag(P, w->D_temp);  orc_mat_postmult_diag(P, w->D_temp);
    orc_mat_premult_diag(A, w->E_temp);  orc_mat_postmult_diag(A, w->D_temp);
    orc_vec_ew_prod(w->D_temp, q, q, n);
    orc_vec_ew_prod(s->D, w->D_temp, s->D, n);
    orc_vec_ew_prod(s->E, w->E_temp, s->E, m);

    /* cost normalisation (scaling.c:113-142) */
    orc_mat_inf_norm_cols_sym_triu(P, w->D_temp);
    c_float c_t = orc_vec_mean(w->D_temp, n);
    c_float qn  = orc_vec_norm_inf(q, n);
    clamp_scaling(&qn, 1);
    if (qn > c_t) c_t = qn;
    clamp_scaling(&c_t, 1);
    c_t = 1. / c_t;
    orc_mat_scale(P, c_t);
    orc_vec_scale(q, c_t, n);
    s->c *= c_t;
  }

  s->cinv = 1. / s->c;
  for (c_int j = 0; j < n; j++) s->Dinv[j] = (c_float)1.0 / s->D[j];
  for (c_int i = 0; i < m; i++) s->Einv[i] = (c_float)1.0 / s->E[i];
  orc_vec_ew_prod(s->E, w->data->l, w->data->l, m);
  orc_vec_ew_prod(s->E, w->data->u, w->data->u, m);
}

/* scaling.c:160-175 */
void orc_unscale_data(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m;
  OSQPScaling *s = w->scaling;
  orc_mat_scale(w->data->P, s->cinv);
  orc_mat_premult_diag(w->data->P, s->Dinv);
  orc_mat_postmult_diag(w->data->P, s->Dinv);
  orc_vec_scale(w->data->q, s->cinv, n);
  orc_vec_ew_prod(s->Dinv, w->data->q, w->data->q, n);
  orc_mat_premult_diag(w->data->A, s->Einv);
  orc_mat_postmult_diag(w->data->A, s->Dinv);
  orc_vec_ew_prod(s->Einv, w->data->l, w->data->l, m);
  orc_vec_ew_prod(s->Einv, w->data->u, w->data->u, m);
}
