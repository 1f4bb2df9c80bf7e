// batch.hip -- batched MPC-style QP engine (one workgroup per QP); filled in below.
#include <hip/hip_runtime.h>
