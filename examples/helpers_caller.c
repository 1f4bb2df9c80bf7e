/* Plain C caller of the helper symbols libosqp_amd.so exports beside the osqp_* API (include/osqp_amd_helpers.h:
 * the reference's cs.h / lin_alg.h / kkt.h routines and the allocator hook), plus -- with the argument "gpu" -- the
 * data-update sequence of the reference's tests/basic_qp/test_basic_qp.h:461-568 through osqp_setup / osqp_update_* /
 * osqp_solve.  Written the way a translation unit of the reference's test-suite uses those symbols.
 *
 *   helpers_caller <problem file> [gpu]
 *
 * The problem file (osqp_amd_read_problem) carries P (upper triangle), A, q, l, u.  Every result is printed as
 * "key: values" with 17 significant digits; tests/test_helpers_c.py compiles this file, runs it on the reference's
 * fixtures (tests/golden) and compares with scipy and the fixtures' expected values. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "osqp_amd.h"

static long n_alloc = 0, n_free = 0;
static void *count_malloc(size_t s) { n_alloc++; return malloc(s); }
static void *count_calloc(size_t k, size_t s) { n_alloc++; return calloc(k, s); }
static void *count_realloc(void *p, size_t s) { if (!p) n_alloc++; return realloc(p, s); }
static void  count_free(void *p) { if (p) n_free++; free(p); }

static void pv(const char *key, const c_float *v, c_int k) {
  printf("%s:", key);
  for (c_int t = 0; t < k; t++) printf(" %.17g", v[t]);
  printf("\n");
}
static void pi(const char *key, const c_int *v, c_int k) {
  printf("%s:", key);
  for (c_int t = 0; t < k; t++) printf(" %lld", (long long)v[t]);
  printf("\n");
}
static void pmat(const char *key, const csc *M, c_int ncols) {
  char name[96];
  snprintf(name, sizeof name, "%s.p", key); pi(name, M->p, ncols + 1);
  snprintf(name, sizeof name, "%s.i", key); pi(name, M->i, M->p[ncols]);
  snprintf(name, sizeof name, "%s.x", key); pv(name, M->x, M->p[ncols]);
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <problem file> [gpu]\n", argv[0]); return 2; }
  osqp_amd_set_allocator(count_malloc, count_calloc, count_realloc, count_free);   /* the leak-counting build of the reference */
  OSQPData *d = NULL;
  if (osqp_amd_read_problem(argv[1], &d)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 3; }
  const c_int n = d->n, m = d->m;
  const csc *P = d->P, *A = d->A;
  c_float *x = d->q, *yv = d->l;          /* an n-vector and an m-vector to multiply with */

  /* ---- lin_alg.h: vectors ---- */
  {
    c_float s[3] = {vec_norm_inf(d->q, n), vec_norm_inf_diff(d->l, d->u, m), vec_scaled_norm_inf(d->q, d->q, n)};
    pv("vec_norms", s, 3);
    c_float t[2] = {vec_mean(d->q, n), vec_prod(d->l, d->u, m)};
    pv("vec_mean_prod", t, 2);
    c_float *c = vec_copy(d->q, n), *e = vec_copy(d->q, n);
    vec_add_scalar(c, 0.5, n); vec_mult_scalar(c, -2.0, n);                 pv("vec_add_mult_scalar", c, n);
    vec_add_scaled(c, d->q, e, n, 3.0);                                      pv("vec_add_scaled", c, n);
    vec_ew_recipr(d->q, c, n);                                               pv("vec_ew_recipr", c, n);
    vec_ew_prod(d->q, e, c, n); vec_ew_sqrt(c, n);                           pv("vec_ew_prod_sqrt", c, n);
    prea_vec_copy(d->q, c, n); vec_ew_max(c, n, 0.1); vec_ew_min(c, n, 0.4); pv("vec_ew_max_min", c, n);
    vec_set_scalar(e, 0.25, n); vec_ew_max_vec(d->q, e, c, n);               pv("vec_ew_max_vec", c, n);
    vec_ew_min_vec(d->q, e, c, n);                                           pv("vec_ew_min_vec", c, n);
    c_int iv[3]; int_vec_set_scalar(iv, 7, 3); c_int iw[3]; prea_int_vec_copy(iv, iw, 3); pi("int_vec", iw, 3);
    c_free(c); c_free(e);
  }
  /* ---- lin_alg.h: matrices ---- */
  {
    c_float *ym = (c_float *)c_malloc((size_t)(m + 1) * sizeof(c_float)), *yn = (c_float *)c_malloc((size_t)n * sizeof(c_float));
    mat_vec(A, x, ym, 0);               pv("mat_vec", ym, m);
    mat_vec(A, x, ym, 1);               pv("mat_vec_pluseq", ym, m);
    mat_vec(A, x, ym, -1); mat_vec(A, x, ym, -1); pv("mat_vec_minuseq", ym, m);
    mat_tpose_vec(A, yv, yn, 0, 0);     pv("mat_tpose_vec", yn, n);
    mat_tpose_vec(A, yv, yn, 1, 0);     pv("mat_tpose_vec_pluseq", yn, n);
    mat_vec(P, x, yn, 0); mat_tpose_vec(P, x, yn, 1, 1);   pv("sym_mat_vec", yn, n);     /* P x from the upper triangle */
    c_float qf = quad_form(P, x);       pv("quad_form", &qf, 1);
    mat_inf_norm_cols(A, yn);           pv("mat_inf_norm_cols", yn, n);
    mat_inf_norm_rows(A, ym);           pv("mat_inf_norm_rows", ym, m);
    mat_inf_norm_cols_sym_triu(P, yn);  pv("mat_inf_norm_cols_sym_triu", yn, n);
    csc *B = copy_csc_mat(A);
    mat_mult_scalar(B, 2.0); mat_premult_diag(B, d->u); mat_postmult_diag(B, d->q);
    pmat("scaled_copy", B, n);
    csc_spfree(B);
    c_free(ym); c_free(yn);
  }
  /* ---- cs.h ---- */
  {
    /* A as a triplet matrix, entries listed last column first; compressing it gives A back column by column */
    const c_int nnz = A->p[n];
    csc *T = csc_spalloc(m, n, nnz, 1, 1);
    c_int *TtoC = (c_int *)c_malloc((size_t)(nnz + 1) * sizeof(c_int));
    c_int z = 0;
    for (c_int j = n - 1; j >= 0; j--)
      for (c_int k = A->p[j]; k < A->p[j + 1]; k++) { T->i[z] = A->i[k]; T->p[z] = j; T->x[z] = A->x[k]; z++; }
    T->nz = z;
    csc *C = triplet_to_csc(T, TtoC);  pmat("triplet_to_csc", C, n);  pi("triplet_to_csc.map", TtoC, nnz);
    csc *R = triplet_to_csr(T, OSQP_NULL); pmat("triplet_to_csr", R, m);
    c_float *D = csc_to_dns(C);        pv("csc_to_dns", D, m * n);
    csc_spfree(C); csc_spfree(R); csc_spfree(T); c_free(TtoC); c_free(D);
    /* full symmetric P -> its upper triangle */
    csc *F = csc_spalloc(n, n, 2 * P->p[n], 1, 1);
    z = 0;
    for (c_int j = 0; j < n; j++)
      for (c_int k = P->p[j]; k < P->p[j + 1]; k++) {
        F->i[z] = P->i[k]; F->p[z] = j; F->x[z] = P->x[k]; z++;
        if (P->i[k] != j) { F->i[z] = j; F->p[z] = P->i[k]; F->x[z] = P->x[k]; z++; }
      }
    F->nz = z;
    csc *Pfull = triplet_to_csc(F, OSQP_NULL);
    csc *U = csc_to_triu(Pfull);       pmat("csc_to_triu", U, n);
    /* symmetric permutation by the reversal */
    c_int *perm = (c_int *)c_malloc((size_t)n * sizeof(c_int));
    for (c_int j = 0; j < n; j++) perm[j] = n - 1 - j;
    c_int *pinv = csc_pinv(perm, n);   pi("csc_pinv", pinv, n);
    c_int *AtoC = (c_int *)c_malloc((size_t)(P->p[n] + 1) * sizeof(c_int));
    csc *S = csc_symperm(P, pinv, AtoC, 1); pmat("csc_symperm", S, n); pi("csc_symperm.map", AtoC, P->p[n]);
    c_int cnt[4] = {3, 0, 2, 5}, cp[5];
    c_int tot = csc_cumsum(cp, cnt, 4); pi("csc_cumsum", cp, 5); pi("csc_cumsum.total", &tot, 1);
    csc_spfree(F); csc_spfree(Pfull); csc_spfree(U); csc_spfree(S); c_free(perm); c_free(pinv); c_free(AtoC);
  }
  /* ---- kkt.h ---- */
  {
    const c_float sigma = 0.5;
    c_float *p2 = (c_float *)c_malloc((size_t)(m + 1) * sizeof(c_float));
    for (c_int i = 0; i < m; i++) p2[i] = 1.0 / (1.6 + 0.1 * (c_float)i);
    c_int *PtoK = (c_int *)c_malloc((size_t)(P->p[n] + 1) * sizeof(c_int)), *AtoK = (c_int *)c_malloc((size_t)(A->p[n] + 1) * sizeof(c_int));
    c_int *rtoK = (c_int *)c_malloc((size_t)(m + 1) * sizeof(c_int)), *Pd = OSQP_NULL, Pdn = 0;
    csc *K = form_KKT(P, A, 0, sigma, p2, PtoK, AtoK, &Pd, &Pdn, rtoK);
    pmat("form_KKT", K, n + m); pi("form_KKT.PtoKKT", PtoK, P->p[n]); pi("form_KKT.AtoKKT", AtoK, A->p[n]);
    pi("form_KKT.param2toKKT", rtoK, m); pi("form_KKT.Pdiag_idx", Pd, Pdn);
    csc *Kr = form_KKT(P, A, 1, sigma, p2, OSQP_NULL, OSQP_NULL, OSQP_NULL, OSQP_NULL, OSQP_NULL);
    pmat("form_KKT_csr", Kr, n + m);
    /* new values: P -> 2 P, A -> -A, param2 -> 3 param2 */
    csc *P2 = copy_csc_mat(P), *A2 = copy_csc_mat(A);
    mat_mult_scalar(P2, 2.0); mat_mult_scalar(A2, -1.0); vec_mult_scalar(p2, 3.0, m);
    update_KKT_P(K, P2, PtoK, sigma, Pd, Pdn); update_KKT_A(K, A2, AtoK); update_KKT_param2(K, p2, rtoK, m);
    pv("update_KKT.x", K->x, K->p[n + m]);
    csc_spfree(K); csc_spfree(Kr); csc_spfree(P2); csc_spfree(A2);
    c_free(p2); c_free(PtoK); c_free(AtoK); c_free(rtoK); c_free(Pd);
  }

  /* ---- the data-update sequence of tests/basic_qp/test_basic_qp.h:461-568 through the API (needs the GPU) ---- */
  if (argc > 2 && !strcmp(argv[2], "gpu")) {
    OSQPSettings *st = (OSQPSettings *)c_malloc(sizeof(OSQPSettings));
    OSQPWorkspace *w = OSQP_NULL;
    osqp_set_default_settings(st);
    st->max_iter = 200; st->alpha = 1.6; st->polish = 1; st->scaling = 0; st->verbose = 0; st->warm_start = 0;
    c_int rc = osqp_setup(&w, d, st);
    if (rc) { printf("setup failed %d\n", (int)rc); return 1; }
    osqp_solve(w);
    printf("solve0: status=%d iter=%d obj=%.12g\n", (int)w->info->status_val, (int)w->info->iter, w->info->obj_val);
    pv("solve0.x", w->solution->x, n); pv("solve0.y", w->solution->y, m);
    /* new linear cost: the workspace's copy must hold it (scaling = 0) */
    c_float *q2 = vec_copy(d->q, n); vec_mult_scalar(q2, 2.5, n); vec_add_scalar(q2, 0.3, n);
    rc = osqp_update_lin_cost(w, q2);
    c_float dq = vec_norm_inf_diff(w->data->q, q2, n);
    printf("update_lin_cost: rc=%d diff=%.3g\n", (int)rc, dq);
    /* new bounds, then a pair with l > u which must be refused */
    c_float *l2 = vec_copy(d->l, m), *u2 = vec_copy(d->u, m);
    vec_add_scalar(l2, -0.2, m); vec_add_scalar(u2, 0.1, m);
    rc = osqp_update_bounds(w, l2, u2);
    printf("update_bounds: rc=%d diff=%.3g %.3g\n", (int)rc, vec_norm_inf_diff(w->data->l, l2, m), vec_norm_inf_diff(w->data->u, u2, m));
    c_float *lbad = vec_copy(u2, m); vec_add_scalar(lbad, 1.0, m);
    printf("update_bounds_bad: rc=%d\n", (int)osqp_update_bounds(w, lbad, u2));
    printf("update_lower_bound_bad: rc=%d\n", (int)osqp_update_lower_bound(w, lbad));
    rc = osqp_update_lower_bound(w, l2);
    printf("update_lower_bound: rc=%d diff=%.3g\n", (int)rc, vec_norm_inf_diff(w->data->l, l2, m));
    rc = osqp_update_upper_bound(w, u2);
    printf("update_upper_bound: rc=%d diff=%.3g\n", (int)rc, vec_norm_inf_diff(w->data->u, u2, m));
    osqp_solve(w);
    printf("solve1: status=%d iter=%d obj=%.12g\n", (int)w->info->status_val, (int)w->info->iter, w->info->obj_val);
    pv("solve1.x", w->solution->x, n); pv("solve1.y", w->solution->y, m);
    pv("solve1.q", q2, n); pv("solve1.l", l2, m); pv("solve1.u", u2, m);
    osqp_cleanup(w);
    c_free(st); c_free(q2); c_free(l2); c_free(u2); c_free(lbad);
  }
  osqp_amd_free_problem(d);
  printf("allocator: allocs=%ld frees=%ld\n", n_alloc, n_free);
  return n_alloc == n_free ? 0 : 4;
}
