/*
 * osqp_amd_helpers.h -- host-side helper symbols of the drop-in boundary.
 *
 * The reference's own callers (tests/ * /test_*.h, examples, language wrappers) use more than
 * osqp.h: the CSC container routines of include/cs.h, the vector / matrix routines of
 * include/lin_alg.h and the KKT assembly of include/kkt.h.  libosqp_amd.so exports them with the
 * reference's names and signatures so that such translation units link against it unchanged
 * (SURVEY.md section 8(b)).  They are plain host C on host arrays -- set-up and test utilities,
 * not part of the device hot path (whose SpMV / reductions live in engine.hip).
 *
 * Each prototype cites the reference declaration it matches (paths relative to /root/reference).
 */
#ifndef OSQP_AMD_HELPERS_H
#define OSQP_AMD_HELPERS_H

#include <stddef.h>
#include "osqp_amd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- memory (include/glob_opts.h:64-73) -----------------------------------------------------
 * The reference selects its allocator at compile time (c_malloc / c_calloc / c_free / c_realloc
 * macros, OSQP_CUSTOM_MEMORY).  A prebuilt shared library offers the same hook at run time:
 * every host allocation of the library (workspace, data copies, plugin object, helper results)
 * goes through these four function pointers; NULL restores the libc function.  Set them before
 * the first osqp_setup and do not change them while workspaces are alive.  (The reference's
 * leak-counting test build, tests/custom_memory/custom_memory.c:7-35, maps onto this.) */
void osqp_amd_set_allocator(void *(*malloc_fn)(size_t), void *(*calloc_fn)(size_t, size_t),
                            void *(*realloc_fn)(void *, size_t), void (*free_fn)(void *));
void *c_malloc(size_t size);
void *c_calloc(size_t num, size_t size);
void *c_realloc(void *ptr, size_t size);
void  c_free(void *ptr);

/* ---- include/cs.h --------------------------------------------------------------------------- */
csc     *csc_matrix(c_int m, c_int n, c_int nzmax, c_float *x, c_int *i, c_int *p);      /* cs.h:26  */
csc     *csc_spalloc(c_int m, c_int n, c_int nzmax, c_int values, c_int triplet);         /* cs.h:44  */
void     csc_spfree(csc *A);                                                               /* cs.h:56  */
csc     *csc_done(csc *C, void *w, void *x, c_int ok);                                     /* cs.h:67  */
csc     *copy_csc_mat(const csc *A);                                                       /* cs.h:80  */
void     prea_copy_csc_mat(const csc *A, csc *B);                                          /* cs.h:86  */
csc     *triplet_to_csc(const csc *T, c_int *TtoC);                                        /* cs.h:105 */
csc     *triplet_to_csr(const csc *T, c_int *TtoC);                                        /* cs.h:119 */
c_float *csc_to_dns(csc *M);                                                               /* cs.h:126 */
csc     *csc_to_triu(csc *M);                                                              /* cs.h:135 */
c_int    csc_cumsum(c_int *p, c_int *c, c_int n);                                          /* cs.h:150 */
c_int   *csc_pinv(c_int const *p, c_int n);                                                /* cs.h:158 */
csc     *csc_symperm(const csc *A, const c_int *pinv, c_int *AtoC, c_int values);          /* cs.h:170 */

/* ---- include/lin_alg.h: vectors ------------------------------------------------------------- */
c_float *vec_copy(c_float *a, c_int n);                                                    /* :17  */
void     prea_vec_copy(const c_float *a, c_float *b, c_int n);                             /* :22  */
void     prea_int_vec_copy(const c_int *a, c_int *b, c_int n);                             /* :27  */
void     vec_set_scalar(c_float *a, c_float sc, c_int n);                                  /* :32  */
void     int_vec_set_scalar(c_int *a, c_int sc, c_int n);                                  /* :37  */
void     vec_add_scalar(c_float *a, c_float sc, c_int n);                                  /* :42  */
void     vec_mult_scalar(c_float *a, c_float sc, c_int n);                                 /* :47  */
void     vec_add_scaled(c_float *c, const c_float *a, const c_float *b, c_int n, c_float sc); /* :52 */
c_float  vec_norm_inf(const c_float *v, c_int l);                                          /* :59  */
c_float  vec_scaled_norm_inf(const c_float *S, const c_float *v, c_int l);                 /* :63  */
c_float  vec_norm_inf_diff(const c_float *a, const c_float *b, c_int l);                   /* :68  */
c_float  vec_mean(const c_float *a, c_int n);                                              /* :73  */
void     vec_ew_recipr(const c_float *a, c_float *b, c_int n);                             /* :79  */
c_float  vec_prod(const c_float *a, const c_float *b, c_int n);                            /* :85  */
void     vec_ew_prod(const c_float *a, const c_float *b, c_float *c, c_int n);             /* :90  */
void     vec_ew_sqrt(c_float *a, c_int n);                                                 /* :98  */
void     vec_ew_max(c_float *a, c_int n, c_float max_val);                                 /* :102 */
void     vec_ew_min(c_float *a, c_int n, c_float min_val);                                 /* :107 */
void     vec_ew_max_vec(const c_float *a, const c_float *b, c_float *c, c_int n);          /* :112 */
void     vec_ew_min_vec(const c_float *a, const c_float *b, c_float *c, c_int n);          /* :118 */

/* ---- include/lin_alg.h: matrices ------------------------------------------------------------ */
void     mat_mult_scalar(csc *A, c_float sc);                                              /* :129 */
void     mat_premult_diag(csc *A, const c_float *d);                                       /* :135 */
void     mat_postmult_diag(csc *A, const c_float *d);                                      /* :141 */
/* y (=, +=, -=) A x for plus_eq 0, 1, -1 */
void     mat_vec(const csc *A, const c_float *x, c_float *y, c_int plus_eq);               /* :150 */
/* y (=, +=, -=) A' x; skip_diag leaves the diagonal out (second half of a symmetric product) */
void     mat_tpose_vec(const csc *A, const c_float *x, c_float *y, c_int plus_eq, c_int skip_diag); /* :162 */
void     mat_inf_norm_cols(const csc *M, c_float *E);                                      /* :177 */
void     mat_inf_norm_rows(const csc *M, c_float *E);                                      /* :186 */
void     mat_inf_norm_cols_sym_triu(const csc *M, c_float *E);                             /* :197 */
c_float  quad_form(const csc *P, const c_float *x);                                        /* :208 */

/* ---- include/kkt.h -------------------------------------------------------------------------- */
/* Upper-triangular KKT matrix [P + param1 I, A'; A, -diag(param2)] in CSC (format 0) or CSR (1),
 * with the optional index maps of kkt.h:44-53 (any of them may be OSQP_NULL). */
csc     *form_KKT(const csc *P, const csc *A, c_int format, c_float param1, c_float *param2,
                  c_int *PtoKKT, c_int *AtoKKT, c_int **Pdiag_idx, c_int *Pdiag_n,
                  c_int *param2toKKT);                                                      /* kkt.h:44 */
void     update_KKT_P(csc *KKT, const csc *P, const c_int *PtoKKT, const c_float param1,
                      const c_int *Pdiag_idx, const c_int Pdiag_n);                        /* kkt.h:69 */
void     update_KKT_A(csc *KKT, const csc *A, const c_int *AtoKKT);                        /* kkt.h:84 */
void     update_KKT_param2(csc *KKT, const c_float *param2, const c_int *param2toKKT,
                           const c_int m);                                                 /* kkt.h:97 */

/* util.h:181-211 (DDEBUG builds of the reference, src/util.c:366-491): printing and dumping of matrices and vectors in the
 * reference's text formats -- dump_csc_matrix writes 1-based "row\tcol\tvalue" triplets closed by "m\tn\t0", dump_vec one
 * "%20.18e" value per line. */
void     print_csc_matrix(csc *M, const char *name);                                       /* util.h:181 */
void     dump_csc_matrix(csc *M, const char *file_name);                                   /* util.h:185 */
void     print_trip_matrix(csc *M, const char *name);                                      /* util.h:189 */
void     print_dns_matrix(c_float *M, c_int m, c_int n, const char *name);                 /* util.h:193 */
void     print_vec(c_float *v, c_int n, const char *name);                                 /* util.h:199 */
void     dump_vec(c_float *v, c_int len, const char *file_name);                           /* util.h:204 */
void     print_vec_int(c_int *x, c_int n, const char *name);                               /* util.h:209 */

#ifdef __cplusplus
}
#endif
#endif
