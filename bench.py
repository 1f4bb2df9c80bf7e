#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native OSQP ADMM engine.

Metric (BASELINE.json): ADMM iterations/s on a single QP, workload = config 2
(random sparse QP, n=10000, m=20000, ~0.1 % nnz, fp64, PCG lin_sys on one
MI355X).  One "step" = one cold-started osqp_solve() of that QP through the C
ABI (problem already set up and resident in HBM); value = ADMM iterations
completed by all ranks / wall time of the timed region.

A single QP does not shard: with --gpus N every rank solves
its own replica (seed = 1 + rank), no data-path collective; scaling is "weak".

Extra objects on the JSON line:
  roofline     dominant PCG kernel: algorithmic bytes per launch / measured
               launch-to-launch period (HIP events on the engine's stream)
  cpu_baseline the oracle (plain-C restatement of the reference CPU path,
               single-threaded direct LDL^T) on a bounded sample of the same
               recipe, timed on this host (rank 0, N=1 only)
  batch        QPs/s of the batched MPC engine (config 4), when built
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RHO0 = 0.1             # reference default (include/constants.h:58)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--m", type=int, default=20000)
    ap.add_argument("--eps", type=float, default=1e-4)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-n", type=int, default=2000, help="size of the bounded CPU sample (m = 2n)")
    ap.add_argument("--batch", type=int, default=1024, help="MPC batch size for the QPs/s leg (0 = skip)")
    ap.add_argument("--no-inexact", action="store_true", help="skip the opt-in inexact-mode leg (shorter traces under rocprofv3)")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 3 / config 5 legs")
    ap.add_argument("--cpu-batch-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-single-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    return ap.parse_args()


def kernel_roofline(solver, reps=300):
    import osqp_amd
    L = osqp_amd.lib()
    L.hipeng_time_kernel.restype = C.c_int
    L.hipeng_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.hipeng_kernel_bytes.restype = C.c_int
    L.hipeng_kernel_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    names = ["k_cg_A", "k_cg_B"]
    rows = []
    for which in range(len(names)):
        us = C.c_double(); by = C.c_double()
        assert L.hipeng_time_kernel(solver.engine(), which, reps, C.byref(us)) == 0
        assert L.hipeng_kernel_bytes(solver.engine(), which, C.byref(by)) == 0
        rows.append(dict(kernel=names[which], usec=us.value, bytes=by.value,
                         gbs=by.value / (us.value * 1e-6) / 1e9))
    pair = C.c_double()
    assert L.hipeng_time_kernel(solver.engine(), 6, reps, C.byref(pair)) == 0       # A then B, as the loop launches them
    n, m = solver.n, solver.m
    # algorithmic bytes of one PCG iteration, B_pcg = 2 S_A + S_P + 8 (10 n + 3 m)
    b_pcg = 2 * (solver.nnzA * 12 + (m + 1) * 4) + (solver.nnzP * 12 + (n + 1) * 4) + 8 * (10 * n + 3 * m)
    per_step = dict(all_kernels=[dict(kernel=r["kernel"], usec=round(r["usec"], 3), gbs=round(r["gbs"], 2)) for r in rows],
                    pcg_iteration=dict(usec=round(pair.value, 3), survey_B_pcg_bytes=b_pcg,
                                       gbs=round(b_pcg / pair.value / 1e3, 2), frac=round(b_pcg / pair.value / 1e3 / HBM_PEAK_GBS, 5),
                                       note="one PCG iteration = k_cg_A + k_cg_B in loop order; bytes = SURVEY 8(d) B_pcg"))
    L.hipeng_resident_info.restype = C.c_int
    L.hipeng_resident_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    info = (C.c_longlong * 16)()
    assert L.hipeng_resident_info(solver.engine(), info) == 0
    if info[1]:
        # Resident engine: ONE launch of k_pcg_resident is the whole linear solve of an ADMM iteration (K in registers,
        # the PCG iterations exchange vectors inside the launch).  Its duration: HIP events around `reps` graph-captured
        # repetitions of [k_pcg_init, k_pcg_resident] on the engine's stream, minus k_pcg_init timed the same way.
        # Units of one launch: the PCG iterations it ran (read back from the device); bytes per unit: SURVEY 8(d) B_pcg.
        reps2 = 100
        # a state in the middle of a solve (60 of its 125 iterations): the repeated linear solve is then a typical one
        # (from the converged iterates the next system needs three PCG iterations)
        solver.update_rho(RHO0)
        solver.update_settings(max_iter=60)
        solver.solve()
        solver.update_settings(max_iter=4000)
        t_pair = C.c_double(); t_init = C.c_double()
        assert L.hipeng_time_kernel(solver.engine(), 8, reps2, C.byref(t_pair)) == 0, "a resident launch of the timed run gave up"
        assert L.hipeng_resident_info(solver.engine(), info) == 0
        its = int(info[6])                       # (k_pcg_init resets the count: read it before timing that kernel alone)
        assert L.hipeng_time_kernel(solver.engine(), 5, reps2, C.byref(t_init)) == 0
        us = t_pair.value - t_init.value
        gbs = b_pcg * its / us / 1e3
        traffic, src, stale = None, "profiles/r03_config2_pmc_and_durations.json", None
        try:
            import hashlib
            prof = json.load(open(os.path.join(ROOT, src)))
            if (n, m) == (10000, 20000):
                k = [v for kk, v in prof["kernels"].items() if kk.startswith("k_pcg_resident")][0]
                traffic = float(k["traffic_bytes_corrected"])
                now = hashlib.sha256(open(os.path.join(ROOT, "osqp_amd", "csrc", "engine.hip"), "rb").read()).hexdigest()
                stale = prof.get("engine_hip_sha256") != now          # the PMC passes ran on another build of engine.hip
        except Exception:
            traffic = None
        return dict(bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 5),
                    traffic=traffic, traffic_from_another_build=stale, kernel="k_pcg_resident",
                    traffic_note="HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE from the committed rocprofv3 PMC passes (%s); "
                                 "the factor 2 on FETCH_SIZE is measured in profiles/r02_fetch_calibration.json; traffic_from_another_build says whether engine.hip has changed since those passes" % src,
                    bytes_per_launch=b_pcg * its, usec_per_launch=round(us, 3), pcg_iterations_per_launch=its,
                    usec_per_pcg_iteration=round(us / max(its, 1), 3),
                    resident=dict(nnzK=int(info[4]), workgroups=int(info[3]), entries_per_thread=int(info[2]), lds_bytes=int(info[5]),
                                  gave_up=int(info[10]), slow_waits=int(info[11]), slow_wait_max_us=round(int(info[12]) * 0.01, 1), republished=int(info[13]),
                                  register_bytes_of_K=int(info[4]) * 10,
                                  # what the launch moves on chip instead of streaming matrices: every CU reads the exchanged vector
                                  # (rows + riding partials, line-padded) once per PCG iteration
                                  exchange_bytes_per_iteration=int(info[3]) * (int(info[5]) - 8 * (512 + 61 + 3 + 16 + 768 + 5 * 64 + 96)),
                                  exchange_gbs_l2_to_cus=round(int(info[3]) * (int(info[5]) - 8 * (512 + 61 + 3 + 16 + 768 + 5 * 64 + 96)) * its / us / 1e3, 1)),
                    note="one launch = one linear solve: K = P + sigma I + A' rho A (%d entries) sits in the register files of %d CUs, "
                         "each PCG iteration exchanges one n-vector between the workgroups inside the launch; duration = (%d graph-captured "
                         "[k_pcg_init, k_pcg_resident] pairs - the same count of k_pcg_init) / %d, every repetition the same solve of %d PCG "
                         "iterations; algorithmic bytes = SURVEY 8(d) B_pcg x iterations (what the launch-per-step kernels stream for the same "
                         "work); the kernel is bound by the latency of the in-launch exchanges (tools/exchange_probe.hip), not by HBM"
                         % (int(info[4]), int(info[3]), reps2, reps2, its),
                    launch_per_step_kernels=per_step)
    dom = max(rows, key=lambda r: r["usec"])
    # HBM-side traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate
    # runs, same workload), corrected as the calibration run prescribes: bytes = 2 x FETCH_SIZE + WRITE_SIZE
    # (profiles/r02_fetch_calibration.json: FETCH_SIZE tallies 128-byte requests at 64 B)
    traffic, src = None, None
    try:
        src = "profiles/r03_config2_pmc_and_durations.json"
        prof = json.load(open(os.path.join(ROOT, src)))
        if (solver.n, solver.m) == (10000, 20000):
            k = [v for kk, v in prof["kernels"].items() if kk.startswith(dom["kernel"])][0]
            traffic = float(k["traffic_bytes_corrected"])
    except Exception:
        traffic = None
    return dict(bound="hbm", achieved=round(dom["gbs"], 2), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(dom["gbs"] / HBM_PEAK_GBS, 5), traffic=traffic, kernel=dom["kernel"],
                traffic_note="HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE from the committed rocprofv3 PMC passes (%s); "
                             "the factor 2 on FETCH_SIZE is measured in profiles/r02_fetch_calibration.json; traffic_from_another_build says whether engine.hip has changed since those passes" % src,
                bytes_per_launch=dom["bytes"], usec_per_launch=round(dom["usec"], 3),
                note="launch-to-launch period of %d graph-captured back-to-back launches (includes the "
                     "dependent-kernel boundary); the 8 MB working set is L2/Infinity-Cache resident" % reps,
                **per_step)


def other_configs(eps):
    """BASELINE configs 3 (Lasso, 7.5 M non-zeros) and 5 (portfolio, 400 dense blocks + a 50 000-entry
    row) at full size: the HBM-bound cases.  One cold solve each (C3 capped at 200 iterations),
    plus the graph-timed PCG kernels with their algorithmic bytes."""
    import osqp_amd
    from osqp_amd.problems import lasso_qp, portfolio_qp
    L = osqp_amd.lib()
    L.hipeng_time_kernel.restype = C.c_int
    L.hipeng_time_kernel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.hipeng_kernel_bytes.restype = C.c_int
    L.hipeng_kernel_bytes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    out = {}
    for name, make, kw in (("config5_portfolio_n50000", lambda: portfolio_qp(), {}),
                           ("config3_lasso_5000x10000", lambda: lasso_qp(), {})):
        full = make()
        pb = {k: v for k, v in full.items() if k in "PqAlu"}
        t0 = time.perf_counter()
        s = osqp_amd.OSQP().setup(**pb, eps_abs=eps, eps_rel=eps, **kw)
        ts = time.perf_counter() - t0
        t0 = time.perf_counter(); r = s.solve(); tv = time.perf_counter() - t0
        st = s.stats()
        rows = []
        # split mode (A dominated by long rows): k_cg_A runs as a vector-update launch (3) and an
        # operator-apply launch (4); the apply carries the matrix stream
        L.hipeng_is_split.restype = C.c_int; L.hipeng_is_split.argtypes = [C.c_void_p]
        split = bool(L.hipeng_is_split(s.engine()))
        for which, nm in (((3, "k_cg_A update-only"), (4, "k_cg_A apply-only")) if split else ((0, "k_cg_A"),)) + ((1, "k_cg_B"),):
            us = C.c_double(); by = C.c_double()
            L.hipeng_time_kernel(s.engine(), which, 100, C.byref(us)); L.hipeng_kernel_bytes(s.engine(), 0 if which in (3, 4) else which, C.byref(by))
            if which == 3: by.value = 8.0 * 12 * s.n          # u,w,p,s,r,Minv,x read + p,s,r,x,u written
            if which == 4: by.value -= 8.0 * 12 * s.n - 8.0 * s.n   # A stream + u gather + rho read, t written
            rows.append(dict(kernel=nm, usec=round(us.value, 2), algorithmic_MB=round(by.value / 1e6, 2),
                             gbs=round(by.value / us.value / 1e3, 1), frac_of_8TBs=round(by.value / us.value / 1e3 / HBM_PEAK_GBS, 4)))
        out[name] = dict(n=int(s.n), m=int(s.m), nnzA=int(s.nnzA), nnzPtriu=int(s.nnzP), setup_s=round(ts, 3),
                         status=r.info.status, admm_iters=int(r.info.iter), solve_s=round(tv, 4),
                         admm_iters_per_s=round(r.info.iter / tv, 1),
                         pcg_iters_per_admm_iter=round(st["pcg_iters_total"] / max(1, r.info.iter), 1),
                         usec_per_pcg_iter=round(1e6 * tv / max(1, st["pcg_iters_total"]), 1), pcg_kernels=rows)
        if st.get("resident"):
            # the linear solves of this config run as resident launches (config 5: k_pcg_blockres -- the dense blocks of P
            # in registers, one exchange of four scalars per workgroup and PCG iteration); the kernels above are what the
            # launch-per-step path (OSQP_AMD_RESIDENT=0) would use
            L.hipeng_resident_info.restype = C.c_int
            L.hipeng_resident_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
            info = (C.c_longlong * 16)()
            s.update_settings(max_iter=100); s.solve(); s.update_settings(max_iter=4000)     # a state in the middle of a solve
            t_pair = C.c_double(); t_init = C.c_double()
            if L.hipeng_time_kernel(s.engine(), 8, 50, C.byref(t_pair)) == 0:
                L.hipeng_resident_info(s.engine(), info)
                L.hipeng_time_kernel(s.engine(), 5, 50, C.byref(t_init))
                us_l = t_pair.value - t_init.value
                out[name]["pcg_kernels_note"] = "launch-per-step kernels, timed for reference; the solve above used resident launches"
                out[name]["resident_launch"] = dict(form={2: "k_pcg_blockres", 3: "k_blk_apply + k_blk_finish (block-direct)",
                                                          4: "k_dd_gather + k_dd_symv_tiles / k_dd_gemv + k_dd_finish (dense-direct)"}.get(int(info[9]), "k_pcg_resident"), usec=round(us_l, 1),
                                                    pcg_iterations=int(info[6]), usec_per_pcg_iteration=round(us_l / max(1, int(info[6])), 2),
                                                    bytes_read_once_per_launch_MB=round((s.nnzP * 2 - s.n) * 8 / 1e6, 1) if info[9] == 2 else None)
                if info[9] == 4:
                    # dense-direct: one pass over the explicit inverse of the dense Schur complement per linear solve
                    nap = int(round(float(info[4]) ** 0.5))
                    # from 2048 unknowns up the inverse is applied from its lower 128 x 128 tiles only (k_dd_symv_tiles): those are the bytes
                    by = 8.0 * float(info[4]) if nap < 2048 else 8.0 * 128 * 128 * (nap // 128) * (nap // 128 + 1) / 2
                    out[name]["resident_launch"].update(dense_unknowns=int(info[3]), sparse_unknowns_by_schur_complement=int(info[15]),
                                                        algorithmic_MB=round(by / 1e6, 1), gbs=round(by / us_l / 1e3, 1), frac_of_8TBs=round(by / us_l / 1e3 / HBM_PEAK_GBS, 4))
                if info[9] == 3:
                    # block-direct: one pass over the inverse blocks (same bytes as the dense blocks of P) per linear solve; the time
                    # includes k_blk_finish (profiles/r03_config5_*: k_blk_apply alone 10.7 us = 4.85 TB/s)
                    by = (s.nnzP * 2 - s.n) * 8.0
                    out[name]["resident_launch"].update(algorithmic_MB=round(by / 1e6, 1), gbs=round(by / us_l / 1e3, 1), frac_of_8TBs=round(by / us_l / 1e3 / HBM_PEAK_GBS, 4))
        if name.startswith("config3"):
            # BASELINE config 3 / SURVEY 8(d): the gamma sweep through osqp_update_lin_cost with warm-started re-solves
            # (docs/examples/lasso.rst:41-63), then perturbed data through osqp_update_A (same pattern) and a warm-started
            # solve (src/osqp.c:1092-1169); update time and iteration rate reported separately, every solve run to `solved`
            from scipy import sparse as _sp
            nf, md = full["n_feat"], full["m_data"]
            sweep = []
            for gamma in (4.0, 7.0, 10.0):
                q = np.concatenate([np.zeros(nf + md), gamma * np.ones(nf)])
                t0 = time.perf_counter(); s.update(q=q); tu = time.perf_counter() - t0
                t0 = time.perf_counter(); rr = s.solve(); tsol = time.perf_counter() - t0
                sweep.append(dict(gamma=gamma, update_ms=round(1e3 * tu, 3), status=rr.info.status, admm_iters=int(rr.info.iter),
                                  solve_s=round(tsol, 4), admm_iters_per_s=round(rr.info.iter / tsol, 1)))
            A = _sp.csc_matrix(pb["A"]); A.sort_indices()
            Ax_new = A.data * (1.0 + 0.01 * np.random.default_rng(0).standard_normal(A.nnz))
            t0 = time.perf_counter(); rcu = s.update(Ax=Ax_new); tu = time.perf_counter() - t0
            t0 = time.perf_counter(); rr = s.solve(); tsol = time.perf_counter() - t0
            out[name]["gamma_sweep_warm_started"] = sweep
            out[name]["update_A_then_warm_solve"] = dict(update_rc=int(rcu), update_A_ms=round(1e3 * tu, 2), nnz_updated=int(A.nnz),
                                                         status=rr.info.status, admm_iters=int(rr.info.iter), solve_s=round(tsol, 4),
                                                         admm_iters_per_s=round(rr.info.iter / tsol, 1),
                                                         note="osqp_update_A: 7.5 M values host -> device, re-equilibration on the device, "
                                                              "mirrors back (src/osqp.c:1092-1169); the solve starts from the previous iterates")
        try:      # the CPU direct solver on the same problem, recorded once by tools/make_config{3,5}_golden.py
            g = json.load(open(os.path.join(ROOT, "tests", "golden", name.split("_")[0] + "_oracle.json")))["info"]
            out[name]["cpu_oracle_recorded"] = dict(admm_iters=g["iters"], admm_iters_per_s=round(g["its"], 2), setup_s=round(g["setup_s"], 1),
                                                    solve_s=round(g["solve_s"], 1), cores=1,
                                                    where="build container, full solve to eps 1e-4 (tests/golden/%s_oracle.json)" % name.split("_")[0])
        except Exception:
            pass
        del s
    out["config5_plus_500_sector_rows"] = config5_sector_rows(eps)
    return out


def config5_sector_rows(eps, rows=500):
    """SURVEY C5 "optionally + 500 sparse sector rows": rows of A that tie variables of different blocks together.  The
    block-direct solve carries every multi-entry row of A as the low-rank term (coupled form); beside it the same
    problem on the launch-per-step PCG kernels (OSQP_AMD_BLOCK_COUPLED_MAX=0), which is where it ran before."""
    import osqp_amd
    from osqp_amd.problems import portfolio_qp
    L = osqp_amd.lib()
    L.hipeng_resident_info.restype = C.c_int
    L.hipeng_resident_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    pb = portfolio_qp(sector_rows=rows)
    res = {}
    ref = None
    for key, cap in (("block_direct_coupled", None), ("launch_per_step_pcg", "0")):
        if cap is not None:
            os.environ["OSQP_AMD_BLOCK_COUPLED_MAX"] = cap
        try:
            t0 = time.perf_counter(); s = osqp_amd.OSQP().setup(**pb, eps_abs=eps, eps_rel=eps); ts = time.perf_counter() - t0
        finally:
            os.environ.pop("OSQP_AMD_BLOCK_COUPLED_MAX", None)
        info = (C.c_longlong * 16)()
        L.hipeng_resident_info(s.engine(), info)
        t0 = time.perf_counter(); r = s.solve(); tv = time.perf_counter() - t0
        st = s.stats()
        t0 = time.perf_counter(); s.update_rho(0.2); tr = time.perf_counter() - t0
        res[key] = dict(setup_s=round(ts, 3), status=r.info.status, admm_iters=int(r.info.iter), solve_s=round(tv, 4),
                        admm_iters_per_s=round(r.info.iter / tv, 1), pcg_iters_per_admm_iter=round(st["pcg_iters_total"] / max(1, r.info.iter), 1),
                        osqp_update_rho_ms=round(1e3 * tr, 2), form=int(info[9]), coupling_rows=int(info[15]))
        if ref is None:
            ref = r
        else:
            res["max_abs_dx_between_the_two"] = float(np.abs(r.x - ref.x).max())
            res["same_iteration_count"] = bool(r.info.iter == ref.info.iter)
        del s
    res["n"], res["m"], res["sector_rows"] = int(pb["P"].shape[0]), int(pb["A"].shape[0]), rows
    return res


def _recorded(cfg):
    """The oracle's full-size run of a config, recorded once in the build container (tests/golden/<cfg>_oracle.json)."""
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", cfg + "_oracle.json")))["info"]
        return dict(admm_iters=g["iters"], admm_iters_per_s=round(g["its"], 3), setup_s=round(g["setup_s"], 1), solve_s=round(g["solve_s"], 1),
                    nnzL=g.get("nnzL"), cores=1, where="build container, the benchmarked instance at full size, solved to eps 1e-4 "
                                                       "(direct LDL^T; fill of the ordering checked in profiles/r02_nnzL_oracle_vs_mmd.txt)")
    except Exception:
        return None


def cpu_single_worker(spec):
    """The single-QP CPU baseline, run in a child process on ONE core while the parent drives the GPU legs (prints one JSON
    object).  Oracle = CPU checker; here it is the timed baseline, never the product path.
      (1) the benchmarked instance itself (n, m of the headline): per-phase times of the direct path's setup
          (qdldl_interface.c:177-323: KKT assembly, ordering, permutation + elimination tree, numeric factorisation); the
          numeric phase takes minutes single-threaded, so it runs for a bounded time with its flops counted and is
          extrapolated by the exact flop count (oracle/orc_ldl.c, orc_ldl_phase_times);
      (2) the same recipe at the bounded sample size, whole solves: the reported `value`."""
    n, m, cpu_n, eps, budget = spec.split(",")
    n, m, cpu_n, eps, budget = int(n), int(m), int(cpu_n), float(eps), float(budget)
    import oracle.oracle as orc
    from osqp_amd import _abi as abi
    from osqp_amd.problems import random_sparse_qp
    from scipy import sparse
    orc.build()
    out = {}
    # ---- (2) bounded sample, whole solves ----
    pb = random_sparse_qp(cpu_n, 2 * cpu_n, seed=1)
    t0 = time.perf_counter()
    s = orc.OracleOSQP().setup(**pb, eps_abs=eps, eps_rel=eps, adaptive_rho_interval=100, warm_start=0)
    t_setup = time.perf_counter() - t0
    iters, t_solve, runs = 0, 0.0, 0
    while t_solve < 10.0 and runs < 20:
        t0 = time.perf_counter()
        s.update_rho(RHO0)           # same protocol as the GPU step: every solve starts from the default rho
        r = s.solve()
        t_solve += time.perf_counter() - t0
        iters += r.info.iter
        runs += 1
    L = orc.lib()
    L.orc_linsys_nnzL.restype = C.c_longlong; L.orc_linsys_nnzL.argtypes = [C.c_void_p]
    nnzL_s = int(L.orc_linsys_nnzL(C.cast(s.work.linsys_solver, C.c_void_p)))
    # seconds per ADMM iteration of the sample without its refactorisations: rho updates excluded by timing iterations alone
    t0 = time.perf_counter(); s.iterate(20); t_iter_s = (time.perf_counter() - t0) / 20
    out["sample"] = dict(n=cpu_n, m=2 * cpu_n, iters=iters, solve_s=t_solve, runs=runs, setup_s=t_setup, nnzL=nnzL_s, iter_s=t_iter_s)
    # ---- (1) the benchmarked instance ----
    pb = random_sparse_qp(n, m, seed=1)
    f = L.orc_ldl_phase_times
    f.restype = C.c_longlong
    f.argtypes = [C.POINTER(abi.csc), C.POINTER(abi.csc), C.c_double, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
    Pu = abi.CscHolder(sparse.triu(pb["P"], format="csc")); Ah = abi.CscHolder(pb["A"])
    rho = np.full(m, RHO0); o = (C.c_double * 8)()
    if f(C.byref(Pu.struct), C.byref(Ah.struct), 1e-6, rho.ctypes.data_as(C.POINTER(C.c_double)), budget, o) == 0:
        out["phases"] = dict(form_kkt_s=o[0], ordering_s=o[1], symbolic_s=o[2], nnzL=int(o[3]), numeric_flops=o[4],
                             numeric_flops_done=o[5], numeric_s_spent=o[6], numeric_finished=bool(o[7]))
    print(json.dumps(out), flush=True)


def cpu_baseline_start(a):
    """Start the single-QP CPU baseline in a child process (before anything in this process touches the GPU)."""
    import subprocess
    spec = "%d,%d,%d,%g,%g" % (a.n, a.m, a.cpu_n, a.eps, 20.0)
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-single-worker", spec], stdout=subprocess.PIPE, text=True)


def cpu_baseline_join(proc, cpu_n):
    try:
        txt = proc.communicate(timeout=240)[0]
        d = json.loads(txt.strip().splitlines()[-1])
    except Exception as ex:                      # the leg is a reported baseline: its failure must not cost the bench line
        try: proc.kill()
        except Exception: pass
        return dict(value=None, unit="ADMM iters/s", cores=1, kind="port", sample="the CPU baseline child failed: %r" % (ex,))
    sm = d["sample"]
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count()
    out = dict(value=round(sm["iters"] / sm["solve_s"], 2), unit="ADMM iters/s", cores=1, kind="port",
               sample="same recipe at n=%d, m=%d (the direct factorisation of the benchmarked n=10000 instance alone takes minutes on one "
                      "core: see benchmarked_instance); %d solves, each osqp_update_rho(0.1) + cold-started osqp_solve (so the in-solve rho "
                      "update and its re-factorisation are included, as on the GPU); setup (ordering + factor) %.2f s excluded; timed in a "
                      "child process on one core of this host while the GPU legs ran" % (sm["n"], sm["m"], sm["runs"], sm["setup_s"]),
               setup_s=round(sm["setup_s"], 3), host_cpus=os.cpu_count(), host_cpus_usable=avail, recorded_full_size=_recorded("config2"))
    ph = d.get("phases")
    if ph:
        rate = ph["numeric_flops_done"] / max(ph["numeric_s_spent"], 1e-9)
        num_s = ph["numeric_s_spent"] if ph["numeric_finished"] else ph["numeric_flops"] / max(rate, 1.0)
        iter_s = sm["iter_s"] * ph["nnzL"] / max(sm["nnzL"], 1)          # triangular solves are linear in nnz(L)
        # one step of the benchmark on this instance: 125 iterations and one re-factorisation (the rho update at iteration 100)
        est = 125.0 / (125.0 * iter_s + num_s)
        out["benchmarked_instance"] = dict(
            n=None, form_kkt_s=round(ph["form_kkt_s"], 3), ordering_s=round(ph["ordering_s"], 2), symbolic_s=round(ph["symbolic_s"], 2), nnzL=ph["nnzL"],
            numeric_gflop=round(ph["numeric_flops"] / 1e9, 1), numeric_gflops_rate_measured=round(rate / 1e9, 2),
            numeric_fraction_measured=round(ph["numeric_flops_done"] / max(ph["numeric_flops"], 1.0), 4), numeric_s_measured=round(ph["numeric_s_spent"], 1),
            numeric_s_extrapolated=round(num_s, 1), iteration_s_extrapolated=round(iter_s, 3), admm_iters_per_s_estimated=round(est, 3),
            note="measured on this host, one core, on the benchmarked instance: assembly, ordering and elimination tree run to the end; the numeric "
                 "factorisation ran for the stated seconds with its flops counted and is extrapolated by the exact flop count sum Lnz (Lnz - 1) "
                 "(an optimistic figure: the first rows have the shortest columns); a solve iteration is extrapolated from the sample by nnz(L); "
                 "estimate = 125 iterations / (125 iteration times + one re-factorisation), the step the GPU is timed on; recorded_full_size is the "
                 "complete run of the same instance in the build container")
        out["benchmarked_instance"].pop("n")
    return out


def rowpart_config5(dist, rank, world, dev_id, eps):
    """BASELINE config 5 "1 -> N MI355X": ONE portfolio QP whose rows of A and blocks of P are sharded over the N ranks
    (include/osqp_amd_rowpart.h: one n-vector all-reduce per PCG iteration, loop in C), next to the same QP on one GPU (rank 0).  Only run
    with N > 1.  Status per DESIGN.md section 7: bound by all-reduce latency; it cannot beat one GPU's block-direct solve
    at this size -- the leg exists so that a multi-GPU node measures that curve instead of leaving it to an estimate."""
    import torch
    from osqp_amd import rowpart
    from osqp_amd.problems import portfolio_qp
    import osqp_amd
    pb = portfolio_qp()
    kw = dict(eps_abs=eps, eps_rel=eps, adaptive_rho_interval=100)
    scaled = rowpart.scaled_problem_from_engine(**pb, device=dev_id)
    # the loop driven from C (osqp_amd_rp_solve, include/osqp_amd_rowpart.h): with the nccl backend its collectives are
    # ncclAllReduce on the engine's stream, with gloo (rehearsal) the group serves as a callback
    rp = rowpart.NativeRowPartitionedOSQP(collective="auto").setup(scaled, device=dev_id, **kw)
    driver = "osqp_amd_rp_solve (C), collective: " + rp.collective
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = rp.solve()
    torch.cuda.synchronize(); dist.barrier()
    dt = time.perf_counter() - t0
    out = None
    if rank == 0:
        s1 = osqp_amd.OSQP().setup(**pb, **kw, warm_start=0)
        s1.solve(); s1.update_rho(RHO0)
        t1 = time.perf_counter(); r1 = s1.solve(); d1 = time.perf_counter() - t1
        out = dict(workload="config5 portfolio n=50000 (400 dense blocks of 125), ONE QP row-partitioned over %d ranks" % world,
                   status=r.info.status, admm_iters=int(r.info.iter), solve_s=round(dt, 4), admm_iters_per_s=round(r.info.iter / dt, 1),
                   pcg_iters=int(r.info.pcg_iters), collectives=int(r.info.collectives), driver=driver,
                   single_gpu=dict(status=r1.info.status, admm_iters=int(r1.info.iter), admm_iters_per_s=round(r1.info.iter / d1, 1)),
                   note="rows of A and blocks of P sharded, n-vectors replicated, one n-vector all-reduce per PCG iteration; the single-GPU figure "
                        "is the block-direct solve (DESIGN.md 2b)")
    return out


def bench_batch(batch, dist, rank, local_rank, world, coll_dev, steps=10, with_cpu=True):
    """Config 4: `batch` MPC QPs (n=120, m=240), contiguous shards of batch/world per
    rank, no communication during the solves, one all_gather of the records."""
    import torch
    import osqp_amd
    from osqp_amd.problems import mpc_batch
    from osqp_amd.dist import shard_range
    s, Q, L, U = mpc_batch(batch)
    lo, hi, per = shard_range(batch, rank, world)
    bs = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q[lo:hi], L[lo:hi], U[lo:hi], device=local_rank,
                                    warm_start=0)
    bs.solve(fetch=False)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bs.solve(fetch=False)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if dist is not None:      # the one collective of the batch path: gather of the per-QP records, device to device
        from osqp_amd.dist import gather_batch_records
        tg = time.perf_counter()
        out = gather_batch_records(bs, per, world, coll_dev)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - tg)
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        rec = out.cpu().numpy()[:batch]
    else:
        r = bs.results()
        rec = np.concatenate([r.x, r.y, r.info_raw], axis=1)
        gather_ms = 0.0
    status = rec[:, s["n"] + s["m"] + 1]
    iters = rec[:, s["n"] + s["m"]]
    # roofline of the batch kernel: it lives in LDS (one workgroup per QP, nothing leaves the CU in the loop), so the
    # bound is LDS bandwidth.  Per ADMM iteration a QP reads/writes in LDS: the K^-1 GEMV input (n), two sparse passes over
    # A (values + patterns + gathered vectors: ~4 words per entry), the z/y/w/x updates (~12 vectors of n or m) and the
    # exchange buffers: bytes/iter = 8 * (4 * 2 * nnzA + 12 * (n + m)) (+ the register-resident K^-1: 0 LDS bytes).
    nnzA_b = int(s["A"].nnz)
    lds_bytes_iter = 8.0 * (4 * 2 * nnzA_b + 12 * (s["n"] + s["m"]))
    tot_iters = float(iters.sum())
    lds_peak = 256 * 128 * 2.4e9 / 1e9        # 256 CUs x 128 B/clk x 2.4 GHz = 78.6 TB/s (MI355X_MICROARCH.md: LDS 64-256 B/clk by instruction; 128 B/clk for b64 accesses)
    flops_iter = 2.0 * s["n"] * s["n"] + 4.0 * 2 * nnzA_b + 20.0 * (s["n"] + s["m"])
    out = dict(metric="QPs/sec (batch)", value=round(batch / dt, 1), unit="QPs/s", batch=batch,
               n=s["n"], m=s["m"], ms_per_batch=round(1e3 * dt, 4), solved=int((status == 1).sum()),
               mean_iters=round(float(iters.mean()), 2), max_iters=int(iters.max()),
               gather_ms=round(gather_ms, 3),
               roofline=dict(bound="lds", achieved=round(lds_bytes_iter * tot_iters / dt / 1e9, 1), peak=round(lds_peak, 1), unit="GB/s",
                             frac=round(lds_bytes_iter * tot_iters / dt / 1e9 / lds_peak, 4), traffic=None,
                             fp64_gflops=round(flops_iter * tot_iters / dt / 1e9, 1),
                             note="LDS-resident kernel: algorithmic LDS bytes per ADMM iteration per QP = 8 (8 nnzA + 12 (n + m)) "
                                  "x iterations of all QPs / batch time, against 256 CUs x 128 B/clk x 2.4 GHz; the kernel is bound by "
                                  "barrier/LDS latency of one workgroup per CU (slowest QP: DESIGN.md), not by either roofline"),
               note="one 512-thread workgroup per QP; cold-started solves (warm_start=0) on the set-up batch "
                    "(scaled data + K^-1 resident in HBM, like the reference's workspace); results left in HBM")
    if world == 1:
        # MPC-style sequence: the measured state moved a little -> osqp_amd_batch_update (new q, l, u
        # from the host, H2D included) -> warm-started solve from the previous solution
        lo_hi = range(lo, hi)
        V = []
        for k in (1, 2):
            Qk, Lk, Uk = np.zeros_like(Q[lo:hi]), np.zeros_like(L[lo:hi]), np.zeros_like(U[lo:hi])
            for j, b in enumerate(lo_hi):
                rng = np.random.default_rng(b)
                x0 = 0.5 * rng.standard_normal(s["nx"]) + 0.02 * k * np.random.default_rng(10_000 * k + b).standard_normal(s["nx"])
                Qk[j], Lk[j], Uk[j] = s["vectors"](x0)
            V.append((Qk, Lk, Uk))
        bw = osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q[lo:hi], L[lo:hi], U[lo:hi], device=local_rank, warm_start=1)
        bw.solve(fetch=False)
        for k in range(2):
            bw.update(Q=V[k][0], L=V[k][1], U=V[k][2]); bw.solve(fetch=False)
        torch.cuda.synchronize()
        t_up = t_so = 0.0
        for k in range(steps):
            Qk, Lk, Uk = V[k % 2]
            t0 = time.perf_counter(); bw.update(Q=Qk, L=Lk, U=Uk); torch.cuda.synchronize(); t1 = time.perf_counter()
            bw.solve(fetch=False); torch.cuda.synchronize(); t2 = time.perf_counter()
            t_up += t1 - t0; t_so += t2 - t1
        rw = bw.results()
        out["warm_start"] = dict(value=round(batch * steps / (t_up + t_so), 1), unit="QPs/s",
                                 ms_update=round(1e3 * t_up / steps, 4), ms_solve=round(1e3 * t_so / steps, 4),
                                 mean_iters=round(float(rw.iter.mean()), 2), solved=int((rw.status_val == 1).sum()),
                                 note="per step: new q,l,u for every QP from the host (state moved by 2 %), "
                                      "osqp_amd_batch_update (H2D + rescale on the device), warm-started solve")
    if with_cpu and rank == 0 and world == 1:
        import oracle.oracle as orc
        orc.build()
        nb = min(batch, 256)
        ws = [orc.OracleOSQP().setup(P=s["P"], q=Q[b], A=s["A"], l=L[b], u=U[b], warm_start=0) for b in range(nb)]
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 5.0:
            for w in ws:
                w.solve()
            reps += 1
        dtc = (time.perf_counter() - t0) / (reps * nb)
        out["cpu_baseline"] = dict(value=round(1.0 / dtc, 1), unit="QPs/s", cores=1, kind="port",
                                   sample="first %d QPs of the batch, solve only (setup excluded), %d passes" % (nb, reps))
        # the same on the host cores this one-GPU job may use (one QP per worker process)
        import subprocess
        try:
            avail = len(os.sched_getaffinity(0))
        except Exception:
            avail = os.cpu_count() or 1
        nw = max(1, min(256, avail, batch))            # every core this job may use, one worker process each
        per_w = max(1, min(batch, 1024) // nw)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-batch-worker",
                                   "%d,%d,%g" % (w * per_w, (w + 1) * per_w, 5.0)], stdout=subprocess.PIPE, text=True)
                 for w in range(nw)]
        tot, tmax = 0, 0.0
        for pr in procs:
            o = pr.communicate()[0].strip().split()
            if pr.returncode == 0 and len(o) == 2:
                tot += int(o[0]); tmax = max(tmax, float(o[1]))
        if tot:
            out["cpu_baseline_all_cores"] = dict(value=round(tot / tmax, 1), unit="QPs/s", cores=nw, kind="port",
                                                 sample="%d worker processes, %d QPs each, solve only, ~5 s" % (nw, per_w))
    return out


def cpu_batch_worker(spec):
    """One CPU worker of the batch baseline: QPs [lo, hi) of the MPC batch, solved round-robin for `secs`."""
    lo, hi, secs = spec.split(",")
    lo, hi, secs = int(lo), int(hi), float(secs)
    import oracle.oracle as orc
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(hi)
    ws = [orc.OracleOSQP().setup(P=s["P"], q=Q[b], A=s["A"], l=L[b], u=U[b], warm_start=0) for b in range(lo, hi)]
    t0 = time.perf_counter(); done = 0
    while time.perf_counter() - t0 < secs:
        for w in ws:
            w.solve()
        done += len(ws)
    print(done, time.perf_counter() - t0)


def main():
    a = parse()
    if a.cpu_batch_worker:
        return cpu_batch_worker(a.cpu_batch_worker)
    if a.cpu_single_worker:
        return cpu_single_worker(a.cpu_single_worker)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: start one process per GPU ourselves (before anything touches the GPU)
        # and relay rank 0's JSON line; the children see WORLD_SIZE and take the branch below
        from osqp_amd.launch import spawn_ranks
        sys.exit(spawn_ranks(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, a.gpus) and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: running %d rank(s)" % (a.gpus, world, world), file=sys.stderr)
    cpu_child = None
    if world == 1 and rank == 0 and not a.no_cpu:
        cpu_child = cpu_baseline_start(a)          # one core, beside the GPU legs; joined before the CPU legs of the batch
    import torch
    dist = None
    ndev = max(1, torch.cuda.device_count())
    dev_id = local_rank % ndev          # == local_rank on a real multi-GPU node
    torch.cuda.set_device(dev_id)
    coll_dev = torch.device("cuda", dev_id) if a.backend == "nccl" else torch.device("cpu")
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):     # BENCH_FORCE_DIST=1: exercise the collectives with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL writes a version banner to stdout when a communicator is created: stdout carries the
        # one JSON line only, so the banner goes to stderr (fd-level, the banner comes from C code)
        sys.stdout.flush()
        saved = os.dup(1); os.dup2(2, 1)
        try:
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=coll_dev)
            else:
                dist.init_process_group(a.backend)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)
    import osqp_amd
    from osqp_amd.problems import random_sparse_qp
    osqp_amd.set_engine_options(device=dev_id)

    pb = random_sparse_qp(a.n, a.m, seed=1 + rank)
    settings = dict(eps_abs=a.eps, eps_rel=a.eps, adaptive_rho_interval=100, warm_start=0, verbose=0)
    solver = osqp_amd.OSQP().setup(**pb, **settings)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        # one step = what a caller of the reference does for this QP after setup: default rho,
        # cold start, osqp_solve (125 iterations with one rho update at iteration 100 on this QP)
        solver.update_rho(RHO0)
        return solver.solve()

    for _ in range(a.warmup):
        ru_w = step()
    barrier()
    t0 = time.perf_counter()
    iters = 0
    last = None
    ru_prev = 0
    for _ in range(a.steps):
        ru_prev = int(last.info.rho_updates) if last is not None else ru_prev
        last = step()
        iters += last.info.iter
    barrier()
    elapsed = time.perf_counter() - t0
    pcg_total = solver.stats()["pcg_iters_total"]
    L_ = osqp_amd.lib()
    L_.hipeng_resident_info.restype = C.c_int; L_.hipeng_resident_info.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    res_buf = (C.c_longlong * 16)(); L_.hipeng_resident_info(solver.engine(), res_buf)
    res_info = [int(v) for v in res_buf]      # [10] launches that gave up, [11] waits over 50 us, [12] the longest (ticks of 10 ns)

    tot_iters, max_t = float(iters), elapsed
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        it = torch.tensor([float(iters)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(it, op=dist.ReduceOp.SUM)
        max_t, tot_iters = float(t.item()), float(it.item())

    batch = None
    if a.batch and world > 1:
        batch = bench_batch(a.batch, dist, rank, dev_id, world, coll_dev, with_cpu=False)
    rowpart5, wedged = None, False
    if world > 1 and not a.no_configs:
        # a reported side leg: it must not cost the headline line.  A rank that fails or stalls in here leaves the others
        # inside a collective, so every rank runs it under an alarm; after a failure the process group is not used again.
        import signal

        def _late(signum, frame):
            raise TimeoutError("row-partitioned leg exceeded its 180 s")
        signal.signal(signal.SIGALRM, _late)
        signal.alarm(180)
        try:
            rowpart5 = rowpart_config5(dist, rank, world, dev_id, a.eps)
        except BaseException as ex:
            wedged = True
            rowpart5 = dict(error=repr(ex)) if rank == 0 else None
        finally:
            signal.alarm(0)

    if rank == 0:
        st = solver.stats()
        out = {
            "metric": "ADMM iters/sec (single QP)", "value": round(tot_iters / max_t, 2),
            "unit": "ADMM iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * max_t / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config2 random sparse QP n=%d m=%d nnzA=%d nnzPtriu=%d, "
                                   "one step = osqp_update_rho(0.1) + cold-started osqp_solve" % (a.n, a.m, solver.nnzA, solver.nnzP),
                       "eps_abs": a.eps, "eps_rel": a.eps, "adaptive_rho_interval": 100,
                       "pcg_eps_rel": osqp_amd.engine_options()["pcg_eps_rel"],
                       "admm_iters_per_solve": int(last.info.iter), "status": last.info.status,
                       "rho_updates_per_solve": int(last.info.rho_updates) - ru_prev if a.steps > 1 else None,
                       "pcg_iters_per_admm_iter": round(pcg_total / max(1, (a.steps + a.warmup) * last.info.iter), 2),
                       "graph_launches": st["graph_launches"], "host_syncs": st["host_syncs"],
                       "resident_gave_up": res_info[10], "resident_slow_waits": res_info[11], "resident_slow_wait_max_us": round(res_info[12] * 0.01, 1),
                       "linear_solves": ("resident launches (K = P + sigma I + A' rho A in registers, one launch per solve)" if st.get("resident")
                                         else "launch-per-step PCG kernels (k_cg_A, k_cg_B)"),
                       "parallelism": "replicas x%d (a single QP does not shard)" % world},
        }
        if world == 1 and not a.no_inexact:
            # opt-in inexact mode (PCG tolerance tied to the ADMM residuals): NOT parity-exact,
            # reported beside the headline, never as `value`
            solver.set_options(pcg_adaptive=1)
            t1 = time.perf_counter(); fi = 0
            for _ in range(a.steps):
                rf = step(); fi += rf.info.iter
            torch.cuda.synchronize()
            tf = time.perf_counter() - t1
            solver.set_options(pcg_adaptive=0)
            out["inexact_mode"] = {"value": round(fi / tf, 2), "unit": "ADMM iters/s", "admm_iters_per_solve": int(rf.info.iter),
                                   "status": rf.info.status, "obj_rel_diff_vs_strict": abs(rf.info.obj_val - last.info.obj_val) / abs(last.info.obj_val),
                                   "note": "OSQP_AMD_PCG_ADAPTIVE=1; results agree with the strict mode only to the ADMM tolerance"}
        if world == 1:
            out["roofline"] = kernel_roofline(solver)
            if not a.no_configs:
                out["other_configs"] = other_configs(a.eps)
            if cpu_child is not None:
                out["cpu_baseline"] = cpu_baseline_join(cpu_child, a.cpu_n)
            if a.batch:                       # (after the single-QP CPU child has finished: its CPU legs use every usable core)
                batch = bench_batch(a.batch, dist, rank, dev_id, world, coll_dev, with_cpu=not a.no_cpu)
        if rowpart5 is not None:
            out["config5_row_partitioned"] = rowpart5
        if batch is not None:
            out["batch"] = batch
        print(json.dumps(out), flush=True)
    if wedged:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
