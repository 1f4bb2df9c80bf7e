"""One-QP-per-stream batches: every `OSQP` workspace owns a HIP stream, so several
independent QPs of any size can be in flight on one GPU.  `osqp_solve` blocks its
calling thread between windows, therefore the solves are driven from a small
thread pool (ctypes drops the GIL inside the C call).  For many *small* QPs with
one sparsity pattern use `BatchOSQP` instead (one workgroup per QP)."""
from concurrent.futures import ThreadPoolExecutor


def solve_many(solvers, max_workers=4):
    """Solve the given set-up `OSQP` objects concurrently; returns their results in order."""
    if len(solvers) <= 1 or max_workers <= 1:
        return [s.solve() for s in solvers]
    with ThreadPoolExecutor(max_workers=min(max_workers, len(solvers))) as ex:
        return list(ex.map(lambda s: s.solve(), solvers))
