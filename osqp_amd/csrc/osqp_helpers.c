/*
 * osqp_helpers.c -- host-side helper symbols of the drop-in boundary (include/osqp_amd_helpers.h):
 * the allocator hook, the CSC container routines (reference include/cs.h), the vector / matrix
 * routines (include/lin_alg.h) and the KKT assembly (include/kkt.h), with the reference's names,
 * signatures and result conventions, so that callers written against the reference link against
 * libosqp_amd.so.  Plain C on host arrays; the device hot path does not come through here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../../include/osqp_amd_helpers.h"

/* ---- allocator hook (glob_opts.h:64-73 as a run-time switch) --------------------------------- */
static void *(*g_malloc)(size_t) = malloc;
static void *(*g_calloc)(size_t, size_t) = calloc;
static void *(*g_realloc)(void *, size_t) = realloc;
static void  (*g_free)(void *) = free;

void osqp_amd_set_allocator(void *(*m)(size_t), void *(*c)(size_t, size_t),
                            void *(*r)(void *, size_t), void (*f)(void *)) {
  g_malloc = m ? m : malloc; g_calloc = c ? c : calloc; g_realloc = r ? r : realloc; g_free = f ? f : free;
}
void *c_malloc(size_t size) { return g_malloc(size); }
void *c_calloc(size_t num, size_t size) { return g_calloc(num, size); }
void *c_realloc(void *ptr, size_t size) { return g_realloc(ptr, size); }
void  c_free(void *ptr) { if (ptr) g_free(ptr); }

#define AT_LEAST_1(v) ((v) > 0 ? (v) : 1)

/* ---- cs.h ------------------------------------------------------------------------------------ */
csc *csc_matrix(c_int m, c_int n, c_int nzmax, c_float *x, c_int *i, c_int *p) {
  csc *M = (csc *)c_malloc(sizeof(csc));
  if (!M) return OSQP_NULL;
  M->m = m; M->n = n; M->nz = -1; M->nzmax = nzmax; M->x = x; M->i = i; M->p = p;
  return M;
}

csc *csc_spalloc(c_int m, c_int n, c_int nzmax, c_int values, c_int triplet) {
  csc *A = (csc *)c_calloc(1, sizeof(csc));
  if (!A) return OSQP_NULL;
  A->m = m; A->n = n;
  A->nzmax = nzmax = AT_LEAST_1(nzmax);
  A->nz = triplet ? 0 : -1;                       /* triplet form counts its entries, compressed form says -1 */
  A->p = (c_int *)c_malloc((size_t)(triplet ? nzmax : n + 1) * sizeof(c_int));
  A->i = (c_int *)c_malloc((size_t)nzmax * sizeof(c_int));
  A->x = values ? (c_float *)c_malloc((size_t)nzmax * sizeof(c_float)) : OSQP_NULL;
  if (!A->p || !A->i || (values && !A->x)) { csc_spfree(A); return OSQP_NULL; }
  return A;
}

void csc_spfree(csc *A) {
  if (!A) return;
  c_free(A->p); c_free(A->i); c_free(A->x); c_free(A);
}

csc *csc_done(csc *C, void *w, void *x, c_int ok) {
  c_free(w); c_free(x);
  if (ok) return C;
  csc_spfree(C);
  return OSQP_NULL;
}

csc *copy_csc_mat(const csc *A) {
  csc *B = csc_spalloc(A->m, A->n, A->p[A->n], 1, 0);
  if (!B) return OSQP_NULL;
  prea_copy_csc_mat(A, B);
  B->nzmax = A->nzmax;
  return B;
}

void prea_copy_csc_mat(const csc *A, csc *B) {
  const c_int nnz = A->p[A->n];
  memcpy(B->p, A->p, (size_t)(A->n + 1) * sizeof(c_int));
  if (nnz > 0) {
    memcpy(B->i, A->i, (size_t)nnz * sizeof(c_int));
    memcpy(B->x, A->x, (size_t)nnz * sizeof(c_float));
  }
  B->nzmax = A->nzmax;
}

c_int csc_cumsum(c_int *p, c_int *c, c_int n) {
  c_int total = 0;
  if (!p || !c) return -1;
  for (c_int k = 0; k < n; k++) { p[k] = total; total += c[k]; c[k] = p[k]; }
  p[n] = total;
  return total;
}

/* triplet -> compressed along `major` (column indices for CSC, row indices for CSR); entries of one
 * compressed column/row keep their triplet order */
static csc *compress(const csc *T, c_int *TtoC, int by_row) {
  const c_int m = T->m, n = T->n, nz = T->nz, dim = by_row ? m : n;
  const c_int *major = by_row ? T->i : T->p, *minor = by_row ? T->p : T->i;
  csc *C = csc_spalloc(m, n, nz, T->x != OSQP_NULL, 0);
  c_int *next = (c_int *)c_calloc((size_t)AT_LEAST_1(dim), sizeof(c_int));
  if (!C || !next) return csc_done(C, next, OSQP_NULL, 0);
  if (by_row) {          /* the compressed pointer array has one entry per row: csc_spalloc sized it for n columns */
    c_free(C->p);
    C->p = (c_int *)c_malloc((size_t)(m + 1) * sizeof(c_int));
    if (!C->p) return csc_done(C, next, OSQP_NULL, 0);
  }
  for (c_int k = 0; k < nz; k++) next[major[k]]++;
  csc_cumsum(C->p, next, dim);
  for (c_int k = 0; k < nz; k++) {
    const c_int dst = next[major[k]]++;
    C->i[dst] = minor[k];
    if (C->x) C->x[dst] = T->x[k];
    if (TtoC) TtoC[k] = dst;
  }
  return csc_done(C, next, OSQP_NULL, 1);
}
csc *triplet_to_csc(const csc *T, c_int *TtoC) { return compress(T, TtoC, 0); }
csc *triplet_to_csr(const csc *T, c_int *TtoC) { return compress(T, TtoC, 1); }

c_float *csc_to_dns(csc *M) {
  c_float *D = (c_float *)c_calloc((size_t)AT_LEAST_1(M->m * M->n), sizeof(c_float));
  if (!D) return OSQP_NULL;
  for (c_int j = 0; j < M->n; j++)
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++) D[j * M->m + M->i[k]] = M->x[k];    /* column major */
  return D;
}

csc *csc_to_triu(csc *M) {
  if (M->m != M->n) return OSQP_NULL;
  c_int cnt = 0;
  for (c_int j = 0; j < M->n; j++)
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++) if (M->i[k] <= j) cnt++;
  csc *U = csc_spalloc(M->m, M->n, cnt, 1, 0);
  if (!U) return OSQP_NULL;
  cnt = 0;
  for (c_int j = 0; j < M->n; j++) {
    U->p[j] = cnt;
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++)
      if (M->i[k] <= j) { U->i[cnt] = M->i[k]; U->x[cnt++] = M->x[k]; }
  }
  U->p[M->n] = cnt;
  U->nzmax = AT_LEAST_1(cnt);
  return U;
}

c_int *csc_pinv(c_int const *p, c_int n) {
  if (!p) return OSQP_NULL;
  c_int *pinv = (c_int *)c_malloc((size_t)AT_LEAST_1(n) * sizeof(c_int));
  if (!pinv) return OSQP_NULL;
  for (c_int k = 0; k < n; k++) pinv[p[k]] = k;
  return pinv;
}

/* C = P A P' for a symmetric A given by its upper triangle; C upper triangular too */
csc *csc_symperm(const csc *A, const c_int *pinv, c_int *AtoC, c_int values) {
  const c_int n = A->n;
  csc *C = csc_spalloc(n, n, A->p[n], values && A->x != OSQP_NULL, 0);
  c_int *cnt = (c_int *)c_calloc((size_t)AT_LEAST_1(n), sizeof(c_int));
  if (!C || !cnt) return csc_done(C, cnt, OSQP_NULL, 0);
  for (c_int j = 0; j < n; j++) {
    const c_int j2 = pinv ? pinv[j] : j;
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      const c_int i = A->i[k];
      if (i > j) continue;                         /* only the upper triangle of A is looked at */
      const c_int i2 = pinv ? pinv[i] : i;
      cnt[i2 > j2 ? i2 : j2]++;
    }
  }
  csc_cumsum(C->p, cnt, n);
  for (c_int j = 0; j < n; j++) {
    const c_int j2 = pinv ? pinv[j] : j;
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      const c_int i = A->i[k];
      if (i > j) continue;
      const c_int i2 = pinv ? pinv[i] : i;
      const c_int dst = cnt[i2 > j2 ? i2 : j2]++;
      C->i[dst] = i2 < j2 ? i2 : j2;
      if (C->x) C->x[dst] = A->x[k];
      if (AtoC) AtoC[k] = dst;
    }
  }
  return csc_done(C, cnt, OSQP_NULL, 1);
}

/* ---- lin_alg.h: vectors ---------------------------------------------------------------------- */
c_float *vec_copy(c_float *a, c_int n) {
  c_float *b = (c_float *)c_malloc((size_t)AT_LEAST_1(n) * sizeof(c_float));
  if (b) prea_vec_copy(a, b, n);
  return b;
}
void prea_vec_copy(const c_float *a, c_float *b, c_int n) { for (c_int i = 0; i < n; i++) b[i] = a[i]; }
void prea_int_vec_copy(const c_int *a, c_int *b, c_int n) { for (c_int i = 0; i < n; i++) b[i] = a[i]; }
void vec_set_scalar(c_float *a, c_float sc, c_int n) { for (c_int i = 0; i < n; i++) a[i] = sc; }
void int_vec_set_scalar(c_int *a, c_int sc, c_int n) { for (c_int i = 0; i < n; i++) a[i] = sc; }
void vec_add_scalar(c_float *a, c_float sc, c_int n) { for (c_int i = 0; i < n; i++) a[i] += sc; }
void vec_mult_scalar(c_float *a, c_float sc, c_int n) { for (c_int i = 0; i < n; i++) a[i] *= sc; }
void vec_add_scaled(c_float *c, const c_float *a, const c_float *b, c_int n, c_float sc) {
  for (c_int i = 0; i < n; i++) c[i] = a[i] + sc * b[i];
}
c_float vec_norm_inf(const c_float *v, c_int l) {
  c_float mx = 0.0;
  for (c_int i = 0; i < l; i++) { const c_float a = fabs(v[i]); if (a > mx) mx = a; }
  return mx;
}
c_float vec_scaled_norm_inf(const c_float *S, const c_float *v, c_int l) {
  c_float mx = 0.0;
  for (c_int i = 0; i < l; i++) { const c_float a = fabs(S[i] * v[i]); if (a > mx) mx = a; }
  return mx;
}
c_float vec_norm_inf_diff(const c_float *a, const c_float *b, c_int l) {
  c_float mx = 0.0;
  for (c_int i = 0; i < l; i++) { const c_float d = fabs(a[i] - b[i]); if (d > mx) mx = d; }
  return mx;
}
c_float vec_mean(const c_float *a, c_int n) {
  c_float s = 0.0;
  for (c_int i = 0; i < n; i++) s += a[i];
  return s / (c_float)n;
}
void vec_ew_recipr(const c_float *a, c_float *b, c_int n) { for (c_int i = 0; i < n; i++) b[i] = (c_float)1.0 / a[i]; }
c_float vec_prod(const c_float *a, const c_float *b, c_int n) {
  c_float s = 0.0;
  for (c_int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}
void vec_ew_prod(const c_float *a, const c_float *b, c_float *c, c_int n) { for (c_int i = 0; i < n; i++) c[i] = b[i] * a[i]; }
void vec_ew_sqrt(c_float *a, c_int n) { for (c_int i = 0; i < n; i++) a[i] = sqrt(a[i]); }
void vec_ew_max(c_float *a, c_int n, c_float max_val) { for (c_int i = 0; i < n; i++) if (a[i] < max_val) a[i] = max_val; }
void vec_ew_min(c_float *a, c_int n, c_float min_val) { for (c_int i = 0; i < n; i++) if (a[i] > min_val) a[i] = min_val; }
void vec_ew_max_vec(const c_float *a, const c_float *b, c_float *c, c_int n) { for (c_int i = 0; i < n; i++) c[i] = a[i] > b[i] ? a[i] : b[i]; }
void vec_ew_min_vec(const c_float *a, const c_float *b, c_float *c, c_int n) { for (c_int i = 0; i < n; i++) c[i] = a[i] < b[i] ? a[i] : b[i]; }

/* ---- lin_alg.h: matrices --------------------------------------------------------------------- */
void mat_mult_scalar(csc *A, c_float sc) {
  const c_int nnz = A->p[A->n];
  for (c_int k = 0; k < nnz; k++) A->x[k] *= sc;
}
void mat_premult_diag(csc *A, const c_float *d) {      /* rows scaled */
  for (c_int j = 0; j < A->n; j++)
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= d[A->i[k]];
}
void mat_postmult_diag(csc *A, const c_float *d) {     /* columns scaled */
  for (c_int j = 0; j < A->n; j++)
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= d[j];
}

void mat_vec(const csc *A, const c_float *x, c_float *y, c_int plus_eq) {
  if (!plus_eq) for (c_int i = 0; i < A->m; i++) y[i] = 0.0;
  if (A->p[A->n] == 0) return;
  for (c_int j = 0; j < A->n; j++)
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      if (plus_eq == -1) y[A->i[k]] -= A->x[k] * x[j];
      else y[A->i[k]] += A->x[k] * x[j];
    }
}

void mat_tpose_vec(const csc *A, const c_float *x, c_float *y, c_int plus_eq, c_int skip_diag) {
  if (!plus_eq) for (c_int j = 0; j < A->n; j++) y[j] = 0.0;
  if (A->p[A->n] == 0) return;
  for (c_int j = 0; j < A->n; j++)
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      const c_int i = A->i[k];
      if (skip_diag && i == j) continue;
      if (plus_eq == -1) y[j] -= A->x[k] * x[i];
      else y[j] += A->x[k] * x[i];
    }
}

void mat_inf_norm_cols(const csc *M, c_float *E) {
  for (c_int j = 0; j < M->n; j++) {
    E[j] = 0.0;
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++) { const c_float a = fabs(M->x[k]); if (a > E[j]) E[j] = a; }
  }
}
void mat_inf_norm_rows(const csc *M, c_float *E) {
  for (c_int i = 0; i < M->m; i++) E[i] = 0.0;
  for (c_int j = 0; j < M->n; j++)
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++) { const c_float a = fabs(M->x[k]); if (a > E[M->i[k]]) E[M->i[k]] = a; }
}
void mat_inf_norm_cols_sym_triu(const csc *M, c_float *E) {
  for (c_int j = 0; j < M->n; j++) E[j] = 0.0;
  for (c_int j = 0; j < M->n; j++)
    for (c_int k = M->p[j]; k < M->p[j + 1]; k++) {
      const c_int i = M->i[k];
      const c_float a = fabs(M->x[k]);
      if (a > E[j]) E[j] = a;
      if (i != j && a > E[i]) E[i] = a;           /* the mirrored entry sits in column i */
    }
}

c_float quad_form(const csc *P, const c_float *x) {     /* 1/2 x'Px from the upper triangle */
  c_float q = 0.0;
  for (c_int j = 0; j < P->n; j++)
    for (c_int k = P->p[j]; k < P->p[j + 1]; k++) {
      const c_int i = P->i[k];
      if (i == j) q += (c_float).5 * P->x[k] * x[i] * x[i];
      else if (i < j) q += P->x[k] * x[i] * x[j];
      else { fprintf(stderr, "ERROR in quad_form: quad_form matrix is not upper triangular\n"); return 0.0; }
    }
  return q;
}

/* ---- kkt.h ----------------------------------------------------------------------------------- */
csc *form_KKT(const csc *P, const csc *A, c_int format, c_float param1, c_float *param2,
              c_int *PtoKKT, c_int *AtoKKT, c_int **Pdiag_idx, c_int *Pdiag_n, c_int *param2toKKT) {
  const c_int n = P->n, m = A->m, N = n + m;
  const c_int cap = P->p[n] + n + A->p[A->n] + m;
  csc *T = csc_spalloc(N, N, cap, 1, 1);
  if (!T) return OSQP_NULL;
  if (Pdiag_idx) { *Pdiag_idx = (c_int *)c_malloc((size_t)AT_LEAST_1(n) * sizeof(c_int)); *Pdiag_n = 0; }
  c_int z = 0;
#define PUT(r, c, v) do { T->i[z] = (r); T->p[z] = (c); T->x[z] = (v); z++; } while (0)
  for (c_int j = 0; j < n; j++) {                 /* (1,1) block: P + param1 I, a diagonal entry in every column */
    c_int has_diag = 0;
    for (c_int k = P->p[j]; k < P->p[j + 1]; k++) {
      const c_int i = P->i[k];
      if (PtoKKT) PtoKKT[k] = z;
      if (i == j) {
        if (Pdiag_idx) (*Pdiag_idx)[(*Pdiag_n)++] = k;
        PUT(i, j, P->x[k] + param1);
        has_diag = 1;
      } else PUT(i, j, P->x[k]);
    }
    if (!has_diag) PUT(j, j, param1);             /* after the column's (strictly upper) entries: rows stay ascending */
  }
  if (Pdiag_idx) *Pdiag_idx = (c_int *)c_realloc(*Pdiag_idx, (size_t)AT_LEAST_1(*Pdiag_n) * sizeof(c_int));
  for (c_int j = 0; j < A->n; j++)                /* (1,2) block: A' */
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      if (AtoKKT) AtoKKT[k] = z;
      PUT(j, n + A->i[k], A->x[k]);
    }
  for (c_int i = 0; i < m; i++) {                 /* (2,2) block: -diag(param2) */
    if (param2toKKT) param2toKKT[i] = z;
    PUT(n + i, n + i, -param2[i]);
  }
#undef PUT
  T->nz = z;
  c_int *map = OSQP_NULL;
  if (PtoKKT || AtoKKT || param2toKKT) {
    map = (c_int *)c_malloc((size_t)AT_LEAST_1(z) * sizeof(c_int));
    if (!map) { csc_spfree(T); if (Pdiag_idx) { c_free(*Pdiag_idx); *Pdiag_idx = OSQP_NULL; } return OSQP_NULL; }
  }
  csc *K = format == 0 ? triplet_to_csc(T, map) : triplet_to_csr(T, map);
  if (K && map) {
    if (PtoKKT) for (c_int k = 0; k < P->p[n]; k++) PtoKKT[k] = map[PtoKKT[k]];
    if (AtoKKT) for (c_int k = 0; k < A->p[A->n]; k++) AtoKKT[k] = map[AtoKKT[k]];
    if (param2toKKT) for (c_int i = 0; i < m; i++) param2toKKT[i] = map[param2toKKT[i]];
  }
  c_free(map);
  csc_spfree(T);
  return K;
}

void update_KKT_P(csc *KKT, const csc *P, const c_int *PtoKKT, const c_float param1,
                  const c_int *Pdiag_idx, const c_int Pdiag_n) {
  const c_int nnz = P->p[P->n];
  for (c_int k = 0; k < nnz; k++) KKT->x[PtoKKT[k]] = P->x[k];
  for (c_int d = 0; d < Pdiag_n; d++) KKT->x[PtoKKT[Pdiag_idx[d]]] += param1;
}
void update_KKT_A(csc *KKT, const csc *A, const c_int *AtoKKT) {
  const c_int nnz = A->p[A->n];
  for (c_int k = 0; k < nnz; k++) KKT->x[AtoKKT[k]] = A->x[k];
}
void update_KKT_param2(csc *KKT, const c_float *param2, const c_int *param2toKKT, const c_int m) {
  for (c_int i = 0; i < m; i++) KKT->x[param2toKKT[i]] = -param2[i];
}


/* ---- debug printing / dumping (the reference builds these under DDEBUG, src/util.c:366-491: same text formats, so that a dump
 *      written here diffs against one written there) ------------------------------------------------------------------------- */
void print_csc_matrix(csc *M, const char *name) {
  printf("%s :\n", name);
  for (c_int j = 0, k = 0; j < M->n; j++)
    for (c_int q = M->p[j]; q < M->p[j + 1]; q++, k++) printf("\t[%3u,%3u] = %.3g\n", (int)M->i[q], (int)j, M->x[k]);
}
void dump_csc_matrix(csc *M, const char *file_name) {          /* 1-based triplets, closed by "m n 0" */
  FILE *f = fopen(file_name, "w");
  if (!f) { fprintf(stderr, "ERROR in %s: Error during writing file %s.\n", __func__, file_name); return; }
  for (c_int j = 0, k = 0; j < M->n; j++)
    for (c_int q = M->p[j]; q < M->p[j + 1]; q++, k++) fprintf(f, "%d\t%d\t%20.18e\n", (int)M->i[q] + 1, (int)j + 1, M->x[k]);
  fprintf(f, "%d\t%d\t%20.18e\n", (int)M->m, (int)M->n, 0.0);
  fclose(f);
  printf("File %s successfully written.\n", file_name);
}
void print_trip_matrix(csc *M, const char *name) {
  printf("%s :\n", name);
  for (c_int k = 0; k < M->nz; k++) printf("\t[%3u, %3u] = %.3g\n", (int)M->i[k], (int)M->p[k], M->x[k]);
}
void print_dns_matrix(c_float *M, c_int m, c_int n, const char *name) {     /* column-major m x n */
  printf("%s : \n\t", name);
  for (c_int i = 0; i < m; i++) {
    for (c_int j = 0; j < n; j++) printf(j < n - 1 ? "% .3g,  " : "% .3g;  ", M[j * m + i]);
    if (i < m - 1) printf("\n\t");
  }
  printf("\n");
}
void print_vec(c_float *v, c_int n, const char *name) { print_dns_matrix(v, 1, n, name); }
void dump_vec(c_float *v, c_int len, const char *file_name) {
  FILE *f = fopen(file_name, "w");
  if (!f) { printf("Error during writing file %s.\n", file_name); return; }
  for (c_int i = 0; i < len; i++) fprintf(f, "%20.18e\n", v[i]);
  fclose(f);
  printf("File %s successfully written.\n", file_name);
}
void print_vec_int(c_int *x, c_int n, const char *name) {
  printf("%s = [", name);
  for (c_int i = 0; i < n; i++) printf(" %i ", (int)x[i]);
  printf("]\n");
}
