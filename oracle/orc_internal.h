/* orc_internal.h -- CPU ORACLE internals (test infrastructure, not the product). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "orc_osqp.h"

c_float  orc_vec_dot(const c_float *a, const c_float *b, c_int n);
c_float  orc_vec_mean(const c_float *a, c_int n);
void     orc_vec_fill(c_float *a, c_float v, c_int n);
void     orc_vec_ew_prod(const c_float *a, const c_float *b, c_float *c, c_int n);
void     orc_vec_scale(c_float *a, c_float s, c_int n);
c_float *orc_vec_dup(const c_float *a, c_int n);
void     orc_mat_scale(csc *A, c_float s);
void     orc_mat_premult_diag(csc *A, const c_float *d);
void     orc_mat_postmult_diag(csc *A, const c_float *d);
csc     *orc_triplet_to_csc(const csc *T, c_int *map);
csc     *orc_symperm_triu(const csc *A, const c_int *pinv, c_int *AtoC);
void     orc_scale_data(OSQPWorkspace *w);
void     orc_unscale_data(OSQPWorkspace *w);
c_int    orc_polish(OSQPWorkspace *w);
double   orc_now(void);

struct OSQP_TIMER { double t0; };
#endif
