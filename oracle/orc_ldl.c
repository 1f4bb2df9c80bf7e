/*
 * orc_ldl.c -- CPU ORACLE (test infrastructure, not the product).
 *
 * Direct KKT plugin of the reference restated in plain C:
 *   - KKT assembly                      src/kkt.c:6-177
 *   - fill-reducing ordering + symmetric permutation with index maps
 *                                        lin_sys/direct/qdldl/qdldl_interface.c:106-173
 *   - LDL^T kernel with the QDLDL v0.1.5 call contract (etree / factor / solve);
 *     QDLDL itself is an empty submodule in the reference tree, so the kernel
 *     is restated from its published up-looking algorithm and from the call
 *     sites qdldl_interface.c:59-99, :344, :389-391, :407-409
 *   - plugin init / solve / update_matrices / update_rho_vec / free
 *                                        qdldl_interface.c:17-43, 177-410
 * The reference orders with SuiteSparse AMD (vendored, but it includes the
 * cmake-generated osqp_configure.h, so it cannot be compiled here); the
 * ordering below is an own approximate-minimum-degree on the quotient graph.
 * Any permutation gives the same solution up to round-off.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc_osqp.h"
#include "orc_internal.h"

/* ======================================================================== */
/* KKT = [P + param1 I, A'; A, -diag(param2)], upper triangle, CSC          */
/* Column layout follows kkt.c:47-122 so rows inside a column are ascending */
/* ======================================================================== */
csc *orc_form_KKT(const csc *P, const csc *A, c_float param1, const c_float *param2,
                  c_int *PtoKKT, c_int *AtoKKT, c_int *param2toKKT) {
  c_int n = P->n, m = A->m;
  c_int cap = P->p[n] + n + A->p[A->n] + m;
  csc *T = orc_csc_alloc(n + m, n + m, cap, 1, 1);
  if (!T) return NULL;
  c_int z = 0;
  for (c_int j = 0; j < n; j++) {
    c_int has_diag = 0;
    for (c_int k = P->p[j]; k < P->p[j + 1]; k++) {
      c_int i = P->i[k];
      T->i[z] = i; T->p[z] = j; T->x[z] = P->x[k];
      if (i == j) { T->x[z] += param1; has_diag = 1; }
      if (PtoKKT) PtoKKT[k] = z;
      z++;
    }
    if (!has_diag) { T->i[z] = j; T->p[z] = j; T->x[z] = param1; z++; }
  }
  for (c_int j = 0; j < A->n; j++)
    for (c_int k = A->p[j]; k < A->p[j + 1]; k++) {
      T->p[z] = n + A->i[k]; T->i[z] = j; T->x[z] = A->x[k];
      if (AtoKKT) AtoKKT[k] = z;
      z++;
    }
  for (c_int i = 0; i < m; i++) {
    T->i[z] = n + i; T->p[z] = n + i; T->x[z] = -param2[i];
    if (param2toKKT) param2toKKT[i] = z;
    z++;
  }
  T->nz = z;
  c_int *map = (c_int *)malloc((size_t)(z > 0 ? z : 1) * sizeof(c_int));
  csc *K = map ? orc_triplet_to_csc(T, map) : NULL;
  if (K) {
    if (PtoKKT) for (c_int k = 0; k < P->p[n]; k++) PtoKKT[k] = map[PtoKKT[k]];
    if (AtoKKT) for (c_int k = 0; k < A->p[A->n]; k++) AtoKKT[k] = map[AtoKKT[k]];
    if (param2toKKT) for (c_int i = 0; i < m; i++) param2toKKT[i] = map[param2toKKT[i]];
  }
  free(map);
  orc_csc_free(T);
  return K;
}

/* ======================================================================== */
/* Approximate minimum degree on the quotient graph                          */
/* ======================================================================== */
typedef struct { c_int *v; c_int len, cap; } ilist;

static int il_push(ilist *l, c_int x) {
  if (l->len == l->cap) {
    c_int nc = l->cap ? 2 * l->cap : 4;
    c_int *nv = (c_int *)realloc(l->v, (size_t)nc * sizeof(c_int));
    if (!nv) return -1;
    l->v = nv; l->cap = nc;
  }
  l->v[l->len++] = x;
  return 0;
}
static void il_free(ilist *l) { free(l->v); l->v = NULL; l->len = l->cap = 0; }

/* degree buckets: doubly linked lists threaded through nxt/prv */
typedef struct { c_int *head, *nxt, *prv; } buckets;
static void bk_insert(buckets *b, c_int d, c_int i) {
  b->prv[i] = -1; b->nxt[i] = b->head[d];
  if (b->head[d] >= 0) b->prv[b->head[d]] = i;
  b->head[d] = i;
}
static void bk_remove(buckets *b, c_int d, c_int i) {
  if (b->prv[i] >= 0) b->nxt[b->prv[i]] = b->nxt[i]; else b->head[d] = b->nxt[i];
  if (b->nxt[i] >= 0) b->prv[b->nxt[i]] = b->prv[i];
}

c_int orc_min_degree_order(c_int n, const c_int *Ap, const c_int *Ai, c_int *perm) {
  if (n <= 0) return 0;
  enum { VAR = 0, ELEM = 1, DEAD = 2, DENSE = 3 };
  ilist *adj = (ilist *)calloc((size_t)n, sizeof(ilist));   /* variable nbrs  */
  ilist *els = (ilist *)calloc((size_t)n, sizeof(ilist));   /* element nbrs   */
  ilist *Le  = (ilist *)calloc((size_t)n, sizeof(ilist));   /* element bodies */
  c_int *deg = (c_int *)calloc((size_t)n, sizeof(c_int));
  c_int *st  = (c_int *)calloc((size_t)n, sizeof(c_int));
  c_int *mark = (c_int *)calloc((size_t)n, sizeof(c_int));
  c_int *wst = (c_int *)calloc((size_t)n, sizeof(c_int));
  c_int *w   = (c_int *)calloc((size_t)n, sizeof(c_int));
  buckets bk;
  bk.head = (c_int *)malloc((size_t)(n + 1) * sizeof(c_int));
  bk.nxt  = (c_int *)malloc((size_t)n * sizeof(c_int));
  bk.prv  = (c_int *)malloc((size_t)n * sizeof(c_int));
  c_int status = 0, nordered = 0;
  if (!adj || !els || !Le || !deg || !st || !mark || !wst || !w || !bk.head || !bk.nxt || !bk.prv) {
    status = -1; goto done;
  }
  for (c_int d = 0; d <= n; d++) bk.head[d] = -1;

  /* symmetric pattern without the diagonal */
  for (c_int j = 0; j < n; j++)
    for (c_int k = Ap[j]; k < Ap[j + 1]; k++) {
      c_int i = Ai[k];
      if (i == j) continue;
      if (il_push(&adj[i], j) || il_push(&adj[j], i)) { status = -1; goto done; }
    }
  /* rows that are nearly full are set aside and ordered last */
  c_int dense_cut = (c_int)(10.0 * sqrt((double)n));
  if (dense_cut < 16) dense_cut = 16;
  c_int ndense = 0;
  for (c_int i = 0; i < n; i++) if (adj[i].len > dense_cut) { st[i] = DENSE; ndense++; }
  for (c_int i = 0; i < n; i++) {
    if (st[i] == DENSE) { il_free(&adj[i]); continue; }
    c_int keep = 0;
    for (c_int t = 0; t < adj[i].len; t++)
      if (st[adj[i].v[t]] != DENSE) adj[i].v[keep++] = adj[i].v[t];
    adj[i].len = keep;
    deg[i] = keep;
    bk_insert(&bk, deg[i], i);
  }

  c_int tag = 0, wtag = 0, mind = 0, nlive = n - ndense;
  for (c_int step = 0; step < nlive; step++) {
    while (mind < n && bk.head[mind] < 0) mind++;
    c_int p = bk.head[mind];
    bk_remove(&bk, mind, p);
    perm[nordered++] = p;

    /* pivot element body Lp = adj(p) U bodies of p's elements, minus p */
    tag++;
    mark[p] = tag;
    ilist Lp = {0, 0, 0};
    for (c_int t = 0; t < adj[p].len; t++) {
      c_int v = adj[p].v[t];
      if (st[v] == VAR && mark[v] != tag) { mark[v] = tag; if (il_push(&Lp, v)) { status = -1; goto done; } }
    }
    for (c_int t = 0; t < els[p].len; t++) {
      c_int e = els[p].v[t];
      if (st[e] != ELEM) continue;
      for (c_int s = 0; s < Le[e].len; s++) {
        c_int v = Le[e].v[s];
        if (st[v] == VAR && mark[v] != tag) { mark[v] = tag; if (il_push(&Lp, v)) { status = -1; goto done; } }
      }
      st[e] = DEAD; il_free(&Le[e]);     /* absorbed into p */
    }
    il_free(&adj[p]); il_free(&els[p]);
    st[p] = ELEM;
    Le[p] = Lp;
    c_int lp = Lp.len;

    /* w[e] = |Le \ Lp| for every element seen from Lp */
    wtag++;
    for (c_int t = 0; t < lp; t++) {
      c_int i = Lp.v[t];
      for (c_int s = 0; s < els[i].len; s++) {
        c_int e = els[i].v[s];
        if (st[e] != ELEM) continue;
        if (wst[e] != wtag) { wst[e] = wtag; w[e] = Le[e].len; }
        w[e]--;
      }
    }
    c_int remaining = nlive - step - 1;   /* variables left after this pivot */
    for (c_int t = 0; t < lp; t++) {
      c_int i = Lp.v[t];
      bk_remove(&bk, deg[i], i);
      c_int keep = 0;
      for (c_int s = 0; s < adj[i].len; s++) {     /* edges now covered by p */
        c_int v = adj[i].v[s];
        if (st[v] == VAR && mark[v] != tag) adj[i].v[keep++] = v;
      }
      adj[i].len = keep;
      c_int d = keep + (lp - 1);
      keep = 0;
      for (c_int s = 0; s < els[i].len; s++) {
        c_int e = els[i].v[s];
        if (st[e] != ELEM) continue;
        if (w[e] == 0) { st[e] = DEAD; il_free(&Le[e]); continue; }  /* subset of Lp */
        els[i].v[keep++] = e;
        d += w[e];
      }
      els[i].len = keep;
      if (il_push(&els[i], p)) { status = -1; goto done; }
      c_int bound = deg[i] + (lp - 1);
      if (d > bound) d = bound;
      if (d > remaining - 1) d = remaining - 1;
      if (d < 0) d = 0;
      deg[i] = d;
      bk_insert(&bk, d, i);
      if (d < mind) mind = d;
    }
  }
  for (c_int i = 0; i < n; i++) if (st[i] == DENSE) perm[nordered++] = i;
  if (nordered != n) status = -2;

done:
  if (adj) for (c_int i = 0; i < n; i++) il_free(&adj[i]);
  if (els) for (c_int i = 0; i < n; i++) il_free(&els[i]);
  if (Le)  for (c_int i = 0; i < n; i++) il_free(&Le[i]);
  free(adj); free(els); free(Le); free(deg); free(st); free(mark); free(wst); free(w);
  free(bk.head); free(bk.nxt); free(bk.prv);
  return status;
}

/* ======================================================================== */
/* LDL^T with the QDLDL call contract                                        */
/* ======================================================================== */
/* Elimination tree + column counts of L.  Returns sum(Lnz), -1 if the input
 * is not upper triangular or a column has no entries, -2 on overflow. */
c_int orc_ldl_etree(c_int n, const c_int *Ap, const c_int *Ai, c_int *work,
                    c_int *Lnz, c_int *etree) {
  for (c_int j = 0; j < n; j++) {
    work[j] = 0; Lnz[j] = 0; etree[j] = -1;
    if (Ap[j] == Ap[j + 1]) return -1;
  }
  for (c_int j = 0; j < n; j++) {
    work[j] = j;
    for (c_int k = Ap[j]; k < Ap[j + 1]; k++) {
      c_int i = Ai[k];
      if (i > j) return -1;
      while (work[i] != j) {          /* climb until a node already tagged j */
        if (etree[i] == -1) etree[i] = j;
        Lnz[i]++;
        work[i] = j;
        i = etree[i];
      }
    }
  }
  c_int total = 0;
  for (c_int j = 0; j < n; j++) {
    if (total > 0x7fffffffffffffffLL - Lnz[j]) return -2;
    total += Lnz[j];
  }
  return total;
}

/* Up-looking numeric factorisation: row k of L is the solution of a sparse
 * triangular system whose pattern is the etree reach of column k of A.
 * Returns the number of positive pivots, or -1 on a zero pivot.
 * iwork holds 3n ints, bwork n bytes, fwork n floats (qdldl_interface.c:256-258). */
c_int orc_ldl_factor(c_int n, const c_int *Ap, const c_int *Ai, const c_float *Ax,
                     c_int *Lp, c_int *Li, c_float *Lx, c_float *D, c_float *Dinv,
                     const c_int *Lnz, const c_int *etree, unsigned char *bwork,
                     c_int *iwork, c_float *fwork) {
  c_int *reach = iwork, *path = iwork + n, *fill = iwork + 2 * n;
  unsigned char *seen = bwork;
  c_float *y = fwork;
  c_int npos = 0;
  Lp[0] = 0;
  for (c_int j = 0; j < n; j++) {
    Lp[j + 1] = Lp[j] + Lnz[j];
    seen[j] = 0; y[j] = 0.0; D[j] = 0.0; fill[j] = Lp[j];
  }
  for (c_int k = 0; k < n; k++) {
    c_int nreach = 0;
    for (c_int t = Ap[k]; t < Ap[k + 1]; t++) {
      c_int i = Ai[t];
      if (i == k) { D[k] = Ax[t]; continue; }
      y[i] = Ax[t];
      if (seen[i]) continue;
      /* walk up the etree from i; push the new path in reverse so that
       * reach[] ends up in a topological (descending-dependency) order */
      c_int plen = 0;
      for (c_int v = i; v != -1 && v < k && !seen[v]; v = etree[v]) {
        seen[v] = 1; path[plen++] = v;
      }
      while (plen) reach[nreach++] = path[--plen];
    }
    for (c_int t = nreach - 1; t >= 0; t--) {
      c_int c = reach[t];
      c_float yc = y[c];
      c_int end = fill[c];
      for (c_int s = Lp[c]; s < end; s++) y[Li[s]] -= Lx[s] * yc;
      Li[end] = k;
      Lx[end] = yc * Dinv[c];
      D[k] -= yc * Lx[end];
      fill[c]++;
      y[c] = 0.0; seen[c] = 0;
    }
    if (D[k] == 0.0) return -1;
    if (D[k] > 0.0) npos++;
    Dinv[k] = 1.0 / D[k];
  }
  return npos;
}

/* x <- (L D L')^{-1} x, L unit lower triangular stored by columns */
void orc_ldl_solve(c_int n, const c_int *Lp, const c_int *Li, const c_float *Lx,
                   const c_float *Dinv, c_float *x) {
  for (c_int j = 0; j < n; j++) {
    c_float xj = x[j];
    for (c_int s = Lp[j]; s < Lp[j + 1]; s++) x[Li[s]] -= Lx[s] * xj;
  }
  for (c_int j = 0; j < n; j++) x[j] *= Dinv[j];
  for (c_int j = n - 1; j >= 0; j--) {
    c_float acc = x[j];
    for (c_int s = Lp[j]; s < Lp[j + 1]; s++) acc -= Lx[s] * x[Li[s]];
    x[j] = acc;
  }
}

/* ======================================================================== */
/* The plugin object                                                         */
/* ======================================================================== */
typedef struct {
  /* vtable prefix -- must mirror struct linsys_solver */
  enum linsys_solver_type type;
  c_int (*solve)(LinSysSolver *self, c_float *b);
  void  (*free)(LinSysSolver *self);
  c_int (*update_matrices)(LinSysSolver *self, const csc *P, const csc *A);
  c_int (*update_rho_vec)(LinSysSolver *self, const c_float *rho_vec);
  c_int nthreads;
  /* private */
  c_int n, m, polish;
  c_float sigma;
  csc *KKT;                 /* permuted upper triangle */
  c_int *perm;              /* perm[new] = old */
  c_int *PtoKKT, *AtoKKT, *rhotoKKT;
  c_int *Lp, *Li; c_float *Lx, *D, *Dinv;
  c_int *etree, *Lnz, *iwork; unsigned char *bwork; c_float *fwork;
  c_float *rho_inv, *bp, *sol;
  c_int nnzL;
} orc_direct;

static void direct_free(LinSysSolver *self) {
  orc_direct *s = (orc_direct *)self;
  if (!s) return;
  orc_csc_free(s->KKT);
  free(s->perm); free(s->PtoKKT); free(s->AtoKKT); free(s->rhotoKKT);
  free(s->Lp); free(s->Li); free(s->Lx); free(s->D); free(s->Dinv);
  free(s->etree); free(s->Lnz); free(s->iwork); free(s->bwork); free(s->fwork);
  free(s->rho_inv); free(s->bp); free(s->sol);
  free(s);
}

static c_int direct_refactor(orc_direct *s) {
  c_int N = s->n + s->m;
  return orc_ldl_factor(N, s->KKT->p, s->KKT->i, s->KKT->x, s->Lp, s->Li, s->Lx,
                        s->D, s->Dinv, s->Lnz, s->etree, s->bwork, s->iwork, s->fwork);
}

/* qdldl_interface.c:341-376 */
static c_int direct_solve(LinSysSolver *self, c_float *b) {
  orc_direct *s = (orc_direct *)self;
  c_int N = s->n + s->m;
  for (c_int k = 0; k < N; k++) s->bp[k] = b[s->perm[k]];
  orc_ldl_solve(N, s->Lp, s->Li, s->Lx, s->Dinv, s->bp);
  if (s->polish) {
    for (c_int k = 0; k < N; k++) b[s->perm[k]] = s->bp[k];
  } else {
    for (c_int k = 0; k < N; k++) s->sol[s->perm[k]] = s->bp[k];
    for (c_int j = 0; j < s->n; j++) b[j] = s->sol[j];
    for (c_int i = 0; i < s->m; i++) b[s->n + i] += s->rho_inv[i] * s->sol[s->n + i];
  }
  return 0;
}

/* qdldl_interface.c:381-393 with kkt.c:184-212 */
static c_int direct_update_matrices(LinSysSolver *self, const csc *P, const csc *A) {
  orc_direct *s = (orc_direct *)self;
  for (c_int k = 0; k < P->p[P->n]; k++) s->KKT->x[s->PtoKKT[k]] = P->x[k];
  for (c_int j = 0; j < P->n; j++)
    for (c_int k = P->p[j]; k < P->p[j + 1]; k++)
      if (P->i[k] == j) s->KKT->x[s->PtoKKT[k]] += s->sigma;
  for (c_int k = 0; k < A->p[A->n]; k++) s->KKT->x[s->AtoKKT[k]] = A->x[k];
  return direct_refactor(s) < 0;
}

/* qdldl_interface.c:396-410 */
static c_int direct_update_rho(LinSysSolver *self, const c_float *rho_vec) {
  orc_direct *s = (orc_direct *)self;
  for (c_int i = 0; i < s->m; i++) {
    s->rho_inv[i] = 1. / rho_vec[i];
    s->KKT->x[s->rhotoKKT[i]] = -s->rho_inv[i];
  }
  return direct_refactor(s) < 0;
}

c_int orc_linsys_nnzL(const LinSysSolver *self) { return ((const orc_direct *)self)->nnzL; }

/* qdldl_interface.c:177-323 */
c_int orc_init_linsys_solver(LinSysSolver **sp, const csc *P, const csc *A,
                             c_float sigma, const c_float *rho_vec, c_int polish) {
  c_int n = P->n, m = A->m, N = n + m;
  orc_direct *s = (orc_direct *)calloc(1, sizeof(orc_direct));
  *sp = (LinSysSolver *)s;
  if (!s) return OSQP_LINSYS_SOLVER_INIT_ERROR;
  s->type = QDLDL_SOLVER; s->nthreads = 1;
  s->solve = direct_solve; s->free = direct_free;
  s->update_matrices = direct_update_matrices; s->update_rho_vec = direct_update_rho;
  s->n = n; s->m = m; s->sigma = sigma; s->polish = polish;

  size_t NN = (size_t)(N > 0 ? N : 1);
  s->rho_inv = (c_float *)malloc((size_t)(m > 0 ? m : 1) * sizeof(c_float));
  s->perm  = (c_int *)malloc(NN * sizeof(c_int));
  s->Lp    = (c_int *)malloc((NN + 1) * sizeof(c_int));
  s->D     = (c_float *)malloc(NN * sizeof(c_float));
  s->Dinv  = (c_float *)malloc(NN * sizeof(c_float));
  s->etree = (c_int *)malloc(NN * sizeof(c_int));
  s->Lnz   = (c_int *)malloc(NN * sizeof(c_int));
  s->iwork = (c_int *)malloc(3 * NN * sizeof(c_int));
  s->bwork = (unsigned char *)malloc(NN);
  s->fwork = (c_float *)malloc(NN * sizeof(c_float));
  s->bp    = (c_float *)malloc(NN * sizeof(c_float));
  s->sol   = (c_float *)malloc(NN * sizeof(c_float));
  c_int *pinv = (c_int *)malloc(NN * sizeof(c_int));
  csc *K0 = NULL;
  c_int *KtoPK = NULL;
  c_int rc = OSQP_LINSYS_SOLVER_INIT_ERROR;
  if (!s->rho_inv || !s->perm || !s->Lp || !s->D || !s->Dinv || !s->etree || !s->Lnz ||
      !s->iwork || !s->bwork || !s->fwork || !s->bp || !s->sol || !pinv) goto fail;

  if (polish) {
    for (c_int i = 0; i < m; i++) s->rho_inv[i] = sigma;      /* -delta I block */
    K0 = orc_form_KKT(P, A, sigma, s->rho_inv, NULL, NULL, NULL);
  } else {
    s->PtoKKT   = (c_int *)malloc((size_t)(P->p[n] + 1) * sizeof(c_int));
    s->AtoKKT   = (c_int *)malloc((size_t)(A->p[A->n] + 1) * sizeof(c_int));
    s->rhotoKKT = (c_int *)malloc((size_t)(m + 1) * sizeof(c_int));
    if (!s->PtoKKT || !s->AtoKKT || !s->rhotoKKT) goto fail;
    for (c_int i = 0; i < m; i++) s->rho_inv[i] = 1. / rho_vec[i];
    K0 = orc_form_KKT(P, A, sigma, s->rho_inv, s->PtoKKT, s->AtoKKT, s->rhotoKKT);
  }
  if (!K0) goto fail;
  if (orc_min_degree_order(N, K0->p, K0->i, s->perm) < 0) goto fail;
  for (c_int k = 0; k < N; k++) pinv[s->perm[k]] = k;
  KtoPK = (c_int *)malloc((size_t)(K0->p[N] + 1) * sizeof(c_int));
  if (!KtoPK) goto fail;
  s->KKT = orc_symperm_triu(K0, pinv, KtoPK);
  if (!s->KKT) goto fail;
  if (!polish) {
    for (c_int k = 0; k < P->p[n]; k++) s->PtoKKT[k] = KtoPK[s->PtoKKT[k]];
    for (c_int k = 0; k < A->p[A->n]; k++) s->AtoKKT[k] = KtoPK[s->AtoKKT[k]];
    for (c_int i = 0; i < m; i++) s->rhotoKKT[i] = KtoPK[s->rhotoKKT[i]];
  }
  c_int nnzL = orc_ldl_etree(N, s->KKT->p, s->KKT->i, s->iwork, s->Lnz, s->etree);
  if (nnzL < 0) goto fail;
  s->nnzL = nnzL;
  s->Li = (c_int *)malloc((size_t)(nnzL + 1) * sizeof(c_int));
  s->Lx = (c_float *)malloc((size_t)(nnzL + 1) * sizeof(c_float));
  if (!s->Li || !s->Lx) goto fail;
  {
    c_int npos = direct_refactor(s);
    if (npos < 0 || npos < n) { rc = OSQP_NONCVX_ERROR; goto fail; }
  }
  free(pinv); free(KtoPK); orc_csc_free(K0);
  return 0;

fail:
  free(pinv); free(KtoPK); orc_csc_free(K0);
  direct_free((LinSysSolver *)s);
  *sp = NULL;
  return rc;
}

/* ======================================================================== */
/* Phase timings of init_linsys_solver with a time budget (bench.py only)    */
/* ======================================================================== */
/* What the direct path costs on ONE core for a problem whose factorisation takes minutes (the benchmarked config-2
 * instance: qdldl_interface.c:177-323 = form_KKT + ordering + permutation + etree + numeric factor).  The phases up to
 * the elimination tree run to the end; the numeric factorisation -- the same up-looking loop as orc_ldl_factor -- runs for
 * at most `budget_s` seconds while its flops are counted, and is extrapolated by the EXACT flop count of the whole
 * factorisation, sum_c Lnz_c (Lnz_c - 1) (each entry appended to column c follows a pass over the entries already there).
 * out: [0] form_KKT s, [1] ordering s, [2] permutation + etree s, [3] nnz(L), [4] flops of the whole numeric phase,
 *      [5] flops done, [6] seconds they took, [7] 1 = the numeric phase finished within the budget. */
c_int orc_ldl_phase_times(const csc *P, const csc *A, c_float sigma, const c_float *rho_vec, c_float budget_s, c_float out[8]) {
  const c_int n = P->n, m = A->m, N = n + m;
  for (int k = 0; k < 8; k++) out[k] = 0.0;
  c_float *rho_inv = (c_float *)malloc((size_t)(m > 0 ? m : 1) * sizeof(c_float));
  c_int *perm = (c_int *)malloc((size_t)(N + 1) * sizeof(c_int)), *pinv = (c_int *)malloc((size_t)(N + 1) * sizeof(c_int));
  c_int *etree = (c_int *)malloc((size_t)(N + 1) * sizeof(c_int)), *Lnz = (c_int *)malloc((size_t)(N + 1) * sizeof(c_int));
  c_int *iwork = (c_int *)malloc((size_t)(3 * N + 3) * sizeof(c_int)), *Lp = (c_int *)malloc((size_t)(N + 2) * sizeof(c_int));
  c_float *D = (c_float *)malloc((size_t)(N + 1) * sizeof(c_float)), *Dinv = (c_float *)malloc((size_t)(N + 1) * sizeof(c_float));
  c_float *y = (c_float *)calloc((size_t)(N + 1), sizeof(c_float));
  unsigned char *seen = (unsigned char *)calloc((size_t)(N + 1), 1);
  csc *K0 = NULL, *K = NULL; c_int *KtoPK = NULL, *Li = NULL; c_float *Lx = NULL;
  c_int rc = -1;
  if (!rho_inv || !perm || !pinv || !etree || !Lnz || !iwork || !Lp || !D || !Dinv || !y || !seen) goto done;
  for (c_int i = 0; i < m; i++) rho_inv[i] = 1. / rho_vec[i];
  double t0 = orc_now();
  K0 = orc_form_KKT(P, A, sigma, rho_inv, NULL, NULL, NULL);
  if (!K0) goto done;
  out[0] = orc_now() - t0; t0 = orc_now();
  if (orc_min_degree_order(N, K0->p, K0->i, perm) < 0) goto done;
  out[1] = orc_now() - t0; t0 = orc_now();
  for (c_int k = 0; k < N; k++) pinv[perm[k]] = k;
  KtoPK = (c_int *)malloc((size_t)(K0->p[N] + 1) * sizeof(c_int));
  if (!KtoPK) goto done;
  K = orc_symperm_triu(K0, pinv, KtoPK);
  if (!K) goto done;
  const c_int nnzL = orc_ldl_etree(N, K->p, K->i, iwork, Lnz, etree);
  if (nnzL < 0) goto done;
  out[2] = orc_now() - t0; out[3] = (c_float)nnzL;
  double ftot = 0.0;
  for (c_int c = 0; c < N; c++) ftot += (double)Lnz[c] * (double)(Lnz[c] > 0 ? Lnz[c] - 1 : 0);
  out[4] = ftot;
  Li = (c_int *)malloc((size_t)(nnzL + 1) * sizeof(c_int)); Lx = (c_float *)malloc((size_t)(nnzL + 1) * sizeof(c_float));
  if (!Li || !Lx) goto done;
  {
    c_int *reach = iwork, *path = iwork + N, *fill = iwork + 2 * N;
    const c_int *Ap = K->p, *Ai = K->i; const c_float *Ax = K->x;
    Lp[0] = 0;
    for (c_int j = 0; j < N; j++) { Lp[j + 1] = Lp[j] + Lnz[j]; D[j] = 0.0; fill[j] = Lp[j]; }
    double fdone = 0.0; t0 = orc_now();
    c_int k = 0;
    for (; k < N; k++) {
      if ((k & 63) == 0 && orc_now() - t0 > budget_s) break;
      c_int nreach = 0;
      for (c_int t = Ap[k]; t < Ap[k + 1]; t++) {
        c_int i = Ai[t];
        if (i == k) { D[k] = Ax[t]; continue; }
        y[i] = Ax[t];
        if (seen[i]) continue;
        c_int plen = 0;
        for (c_int v = i; v != -1 && v < k && !seen[v]; v = etree[v]) { seen[v] = 1; path[plen++] = v; }
        while (plen) reach[nreach++] = path[--plen];
      }
      for (c_int t = nreach - 1; t >= 0; t--) {
        c_int c = reach[t];
        c_float yc = y[c];
        c_int end = fill[c];
        for (c_int q = Lp[c]; q < end; q++) y[Li[q]] -= Lx[q] * yc;
        fdone += 2.0 * (double)(end - Lp[c]);
        Li[end] = k; Lx[end] = yc * Dinv[c];
        D[k] -= yc * Lx[end];
        fill[c]++; y[c] = 0.0; seen[c] = 0;
      }
      if (D[k] == 0.0) break;
      Dinv[k] = 1.0 / D[k];
    }
    out[5] = fdone; out[6] = orc_now() - t0; out[7] = k == N ? 1.0 : 0.0;
  }
  rc = 0;
done:
  free(rho_inv); free(perm); free(pinv); free(etree); free(Lnz); free(iwork); free(Lp); free(D); free(Dinv); free(y); free(seen);
  free(KtoPK); free(Li); free(Lx); orc_csc_free(K0); orc_csc_free(K);
  return rc;
}
