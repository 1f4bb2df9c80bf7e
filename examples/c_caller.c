/* A plain C caller of libosqp_amd.so, written against include/osqp_amd.h exactly as a
 * program written against the reference's include/osqp.h would be (setup, solve, update
 * the linear cost and the bounds, solve again, cleanup).  Data: the 2-variable /
 * 3-constraint demo QP (reference: examples/osqp_demo.c:6-18).  Prints one line per
 * solve; tests/test_gpu_parity.py compiles, runs and checks it. */
#include <stdio.h>
#include <stdlib.h>
#include "osqp_amd.h"

int main(void) {
  c_float P_x[3] = {4.0, 1.0, 2.0};
  c_int   P_i[3] = {0, 0, 1};
  c_int   P_p[3] = {0, 1, 3};
  c_float q[2]   = {1.0, 1.0};
  c_float A_x[4] = {1.0, 1.0, 1.0, 1.0};
  c_int   A_i[4] = {0, 1, 0, 2};
  c_int   A_p[3] = {0, 2, 4};
  c_float l[3]   = {1.0, 0.0, 0.0};
  c_float u[3]   = {1.0, 0.7, 0.7};
  c_int n = 2, m = 3, rc;

  OSQPSettings *settings = (OSQPSettings *)malloc(sizeof(OSQPSettings));
  OSQPData *data = (OSQPData *)malloc(sizeof(OSQPData));
  OSQPWorkspace *work = NULL;
  if (!settings || !data) return 100;
  data->n = n; data->m = m;
  data->P = csc_matrix(n, n, 3, P_x, P_i, P_p);
  data->q = q;
  data->A = csc_matrix(m, n, 4, A_x, A_i, A_p);
  data->l = l; data->u = u;
  osqp_set_default_settings(settings);
  settings->alpha = 1.0;
  settings->verbose = 0;

  rc = (int)osqp_setup(&work, data, settings);
  if (rc) { printf("setup failed %d\n", (int)rc); return 1; }
  rc = (int)osqp_solve(work);
  printf("solve1 rc=%d status=%d iter=%d obj=%.10f x=%.8f,%.8f y=%.8f,%.8f,%.8f\n", (int)rc,
         (int)work->info->status_val, (int)work->info->iter, work->info->obj_val,
         work->solution->x[0], work->solution->x[1],
         work->solution->y[0], work->solution->y[1], work->solution->y[2]);

  {
    c_float q2[2] = {2.0, 3.0}, l2[3] = {2.0, -1.0, -1.0}, u2[3] = {2.0, 2.5, 2.5};
    if (osqp_update_lin_cost(work, q2) || osqp_update_bounds(work, l2, u2)) { printf("update failed\n"); return 2; }
  }
  rc = (int)osqp_solve(work);
  printf("solve2 rc=%d status=%d iter=%d obj=%.10f x=%.8f,%.8f y=%.8f,%.8f,%.8f\n", (int)rc,
         (int)work->info->status_val, (int)work->info->iter, work->info->obj_val,
         work->solution->x[0], work->solution->x[1],
         work->solution->y[0], work->solution->y[1], work->solution->y[2]);

  osqp_cleanup(work);
  free(data->A); free(data->P); free(data); free(settings);
  return 0;
}
