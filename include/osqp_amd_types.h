/*
 * osqp_amd_types.h -- C ABI data types of the drop-in boundary.
 *
 * These structs are laid out field-for-field like the reference's default
 * desktop build (DLONG, double precision, PROFILING, PRINTING, not EMBEDDED) so
 * that a caller compiled against the reference's headers can hand its
 * OSQPData / OSQPSettings to this library and read OSQPInfo / OSQPSolution back
 * unchanged.  Layout sources (reference, /root/reference):
 *   csc            include/types.h:21-29
 *   OSQPScaling    include/types.h:45-52
 *   OSQPSolution   include/types.h:57-60
 *   OSQPInfo       include/types.h:66-91
 *   OSQPPolish     include/types.h:99-114
 *   OSQPData       include/types.h:125-133
 *   OSQPSettings   include/types.h:139-176
 *   OSQPWorkspace  include/types.h:182-289
 *   linsys vtable  include/types.h:298-319
 *   constants      include/constants.h:14-121
 *   c_int/c_float  include/glob_opts.h:79-90
 */
#ifndef OSQP_AMD_TYPES_H
#define OSQP_AMD_TYPES_H

#ifdef __cplusplus
extern "C" {
#endif

typedef long long c_int;   /* index type at the ABI (reference DLONG default) */
typedef double    c_float; /* value type at the ABI */

/* ---- status values (info->status_val) ---------------------------------- */
#define OSQP_DUAL_INFEASIBLE_INACCURATE   (4)
#define OSQP_PRIMAL_INFEASIBLE_INACCURATE (3)
#define OSQP_SOLVED_INACCURATE            (2)
#define OSQP_SOLVED                       (1)
#define OSQP_MAX_ITER_REACHED             (-2)
#define OSQP_PRIMAL_INFEASIBLE            (-3)
#define OSQP_DUAL_INFEASIBLE              (-4)
#define OSQP_SIGINT                       (-5)
#define OSQP_TIME_LIMIT_REACHED           (-6)
#define OSQP_NON_CVX                      (-7)
#define OSQP_UNSOLVED                     (-10)

/* ---- linear-system solver ids ------------------------------------------
 * 0 and 1 are the reference's ids (constants.h:35).  HIP_PCG_SOLVER is the id
 * this library adds for its device-resident indirect solver; see
 * INTEGRATION.md for the one-line edit on the reference side. */
enum linsys_solver_type {
  QDLDL_SOLVER       = 0,
  MKL_PARDISO_SOLVER = 1,
  HIP_PCG_SOLVER     = 2,
  UNKNOWN_SOLVER     = 99
};

/* ---- setup error codes (constants.h:42-50) ----------------------------- */
enum osqp_error_type {
  OSQP_DATA_VALIDATION_ERROR = 1,
  OSQP_SETTINGS_VALIDATION_ERROR,
  OSQP_LINSYS_SOLVER_LOAD_ERROR,
  OSQP_LINSYS_SOLVER_INIT_ERROR,
  OSQP_NONCVX_ERROR,
  OSQP_MEM_ALLOC_ERROR,
  OSQP_WORKSPACE_NOT_INIT_ERROR
};

/* ---- algorithm constants (constants.h:58-121) -------------------------- */
#define RHO                  (0.1)
#define SIGMA                (1E-06)
#define MAX_ITER             (4000)
#define EPS_ABS              (1E-3)
#define EPS_REL              (1E-3)
#define EPS_PRIM_INF         (1E-4)
#define EPS_DUAL_INF         (1E-4)
#define ALPHA                (1.6)
#define RHO_MIN              (1e-06)
#define RHO_MAX              (1e06)
#define RHO_EQ_OVER_RHO_INEQ (1e03)
#define RHO_TOL              (1e-04)
#define DELTA                (1E-6)
#define POLISH               (0)
#define POLISH_REFINE_ITER   (3)
#define VERBOSE              (1)
#define SCALED_TERMINATION   (0)
#define CHECK_TERMINATION    (25)
#define WARM_START           (1)
#define SCALING              (10)
#define MIN_SCALING          (1e-04)
#define MAX_SCALING          (1e+04)
#define OSQP_NULL            0
/* NB: the reference's "NaN" is the *number* 2143289344.0, not an IEEE NaN */
#define OSQP_NAN             ((c_float)0x7fc00000UL)
#define OSQP_INFTY           ((c_float)1e30)
#define OSQP_DIVISION_TOL    ((c_float)1.0 / OSQP_INFTY)
#define ADAPTIVE_RHO                      (1)
#define ADAPTIVE_RHO_INTERVAL             (0)
#define ADAPTIVE_RHO_FRACTION             (0.4)
#define ADAPTIVE_RHO_MULTIPLE_TERMINATION (4)
#define ADAPTIVE_RHO_FIXED                (100)
#define ADAPTIVE_RHO_TOLERANCE            (5)
#define TIME_LIMIT                        (0)
#define PRINT_INTERVAL                    200

/* ---- sparse matrix, compressed-column (or triplet when nz >= 0) -------- */
typedef struct {
  c_int    nzmax;
  c_int    m;
  c_int    n;
  c_int   *p;
  c_int   *i;
  c_float *x;
  c_int    nz;
} csc;

typedef struct linsys_solver LinSysSolver;
typedef struct OSQP_TIMER OSQPTimer;

typedef struct {
  c_float  c;
  c_float *D;
  c_float *E;
  c_float  cinv;
  c_float *Dinv;
  c_float *Einv;
} OSQPScaling;

typedef struct {
  c_float *x;
  c_float *y;
} OSQPSolution;

typedef struct {
  c_int   iter;
  char    status[32];
  c_int   status_val;
  c_int   status_polish;
  c_float obj_val;
  c_float pri_res;
  c_float dua_res;
  c_float setup_time;
  c_float solve_time;
  c_float update_time;
  c_float polish_time;
  c_float run_time;
  c_int   rho_updates;
  c_float rho_estimate;
} OSQPInfo;

typedef struct {
  csc     *Ared;
  c_int    n_low;
  c_int    n_upp;
  c_int   *A_to_Alow;
  c_int   *A_to_Aupp;
  c_int   *Alow_to_A;
  c_int   *Aupp_to_A;
  c_float *x;
  c_float *z;
  c_float *y;
  c_float  obj_val;
  c_float  pri_res;
  c_float  dua_res;
} OSQPPolish;

typedef struct {
  c_int    n;
  c_int    m;
  csc     *P; /* upper triangle only */
  csc     *A;
  c_float *q;
  c_float *l;
  c_float *u;
} OSQPData;

typedef struct {
  c_float rho;
  c_float sigma;
  c_int   scaling;
  c_int   adaptive_rho;
  c_int   adaptive_rho_interval;
  c_float adaptive_rho_tolerance;
  c_float adaptive_rho_fraction;
  c_int   max_iter;
  c_float eps_abs;
  c_float eps_rel;
  c_float eps_prim_inf;
  c_float eps_dual_inf;
  c_float alpha;
  enum linsys_solver_type linsys_solver;
  c_float delta;
  c_int   polish;
  c_int   polish_refine_iter;
  c_int   verbose;
  c_int   scaled_termination;
  c_int   check_termination;
  c_int   warm_start;
  c_float time_limit;
} OSQPSettings;

typedef struct {
  OSQPData     *data;
  LinSysSolver *linsys_solver;
  OSQPPolish   *pol;
  c_float *rho_vec;
  c_float *rho_inv_vec;
  c_int   *constr_type;
  c_float *x;
  c_float *y;
  c_float *z;
  c_float *xz_tilde;
  c_float *x_prev;
  c_float *z_prev;
  c_float *Ax;
  c_float *Px;
  c_float *Aty;
  c_float *delta_y;
  c_float *Atdelta_y;
  c_float *delta_x;
  c_float *Pdelta_x;
  c_float *Adelta_x;
  c_float *D_temp;
  c_float *D_temp_A;
  c_float *E_temp;
  OSQPSettings *settings;
  OSQPScaling  *scaling;
  OSQPSolution *solution;
  OSQPInfo     *info;
  OSQPTimer    *timer;
  c_int first_run;
  c_int clear_update_time;
  c_int rho_update_from_solve;
  c_int summary_printed;
} OSQPWorkspace;

/* Plugin vtable: every solver struct starts with exactly these members. */
struct linsys_solver {
  enum linsys_solver_type type;
  c_int (*solve)(LinSysSolver *self, c_float *b);
  void  (*free)(LinSysSolver *self);
  c_int (*update_matrices)(LinSysSolver *self, const csc *P, const csc *A);
  c_int (*update_rho_vec)(LinSysSolver *self, const c_float *rho_vec);
  c_int nthreads;
};

#ifdef __cplusplus
}
#endif
#endif /* OSQP_AMD_TYPES_H */
