"""CPU tests of the product side that need no GPU: the C-ABI library loads and
exports every symbol the headers in include/ declare; struct layouts match the
C side; setup fails loudly (no CPU fallback) when no HIP device is present;
the sharded batch path gathers correctly over gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", txt)
    return sorted(set(n for n in names if not n.startswith("__") and n not in ("defined", "int", "void", "double")))      # (a function-pointer typedef reads as "int (...)")


@pytest.fixture(scope="module")
def product_lib():
    import osqp_amd
    osqp_amd.build()
    return osqp_amd.lib()


@pytest.mark.parametrize("header", ["osqp_amd.h", "osqp_amd_engine.h", "osqp_amd_batch.h", "osqp_amd_helpers.h", "osqp_amd_rowpart.h"])
def test_library_exports_every_declared_symbol(product_lib, header):
    names = _declared_functions(header)
    assert len(names) >= 6
    missing = [n for n in names if not hasattr(product_lib, n)]
    assert not missing, missing


def test_struct_sizes_match_c(product_lib, tmp_path):
    """ctypes mirrors (osqp_amd/_abi.py) vs sizeof/offsetof compiled from the header."""
    from osqp_amd import abi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "osqp_amd_types.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(csc), sizeof(OSQPSettings),'
                   'sizeof(OSQPInfo), sizeof(OSQPData), sizeof(OSQPWorkspace), sizeof(struct linsys_solver),'
                   'offsetof(OSQPSettings, linsys_solver), offsetof(OSQPInfo, obj_val));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(t) for t in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.csc), C.sizeof(abi.OSQPSettings), C.sizeof(abi.OSQPInfo), C.sizeof(abi.OSQPData),
            C.sizeof(abi.OSQPWorkspace), C.sizeof(abi.LinSysSolver), abi.OSQPSettings.linsys_solver.offset,
            abi.OSQPInfo.obj_val.offset]
    assert got == want


def test_default_settings_match_reference_constants(product_lib):
    """include/constants.h:58-118 defaults (linsys_solver defaults to the HIP PCG id)."""
    import osqp_amd
    from osqp_amd import abi
    st = abi.OSQPSettings()
    product_lib.osqp_set_default_settings.restype = None
    product_lib.osqp_set_default_settings.argtypes = [C.POINTER(abi.OSQPSettings)]
    product_lib.osqp_set_default_settings(C.byref(st))
    assert (st.rho, st.sigma, st.scaling, st.max_iter) == (0.1, 1e-6, 10, 4000)
    assert (st.eps_abs, st.eps_rel, st.eps_prim_inf, st.eps_dual_inf, st.alpha) == (1e-3, 1e-3, 1e-4, 1e-4, 1.6)
    assert (st.adaptive_rho, st.adaptive_rho_interval, st.adaptive_rho_tolerance, st.adaptive_rho_fraction) == (1, 0, 5.0, 0.4)
    assert (st.check_termination, st.warm_start, st.polish, st.polish_refine_iter, st.delta) == (25, 1, 0, 3, 1e-6)
    assert st.linsys_solver == abi.HIP_PCG_SOLVER


def test_validation_happens_before_any_device_work(product_lib):
    """Invalid data / settings are rejected with the reference's codes 1 / 2 on a CPU-only box."""
    import osqp_amd
    pb, _ = load_golden("basic_qp")
    with pytest.raises(ValueError, match="error 2"):
        osqp_amd.OSQP().setup(**pb, rho=-1.0)
    bad = dict(pb); bad["l"] = pb["u"] + 1.0; bad["l"][3] = 0.0
    with pytest.raises(ValueError, match="error 1"):
        osqp_amd.OSQP().setup(**bad)


def test_no_cpu_fallback_without_gpu(product_lib):
    """Without a HIP device the product refuses to set up (OSQP_LINSYS_SOLVER_LOAD_ERROR = 3)
    instead of silently computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import osqp_amd
    pb, _ = load_golden("basic_qp")
    with pytest.raises(ValueError, match="error 3"):
        osqp_amd.OSQP().setup(**pb)
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(2)
    with pytest.raises(ValueError, match="error 3"):
        osqp_amd.BatchOSQP().setup(s["P"], s["A"], Q, L, U)


def test_product_sources_do_not_reference_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "osqp_amd")):
        for f in fs:
            if f.endswith((".py", ".c", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"\boracle\b|\borc_", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
    for h in os.listdir(os.path.join(ROOT, "include")):
        assert "orc_" not in open(os.path.join(ROOT, "include", h)).read()


def test_shard_ranges_cover_the_batch():
    from osqp_amd.dist import shard_range
    for B in (1, 7, 1024, 1025):
        for W in (1, 2, 3, 8):
            got = []
            for r in range(W):
                lo, hi, per = shard_range(B, r, W)
                got += list(range(lo, hi))
                assert hi - lo <= per
            assert got == list(range(B))


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from osqp_amd.dist import sharded_batch_solve
from osqp_amd.problems import mpc_batch
import oracle.oracle as orc
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
s, Q, L, U = mpc_batch(6)
def local(Qs, Ls, Us):      # CPU stand-in for the per-rank GPU batch solve (test only)
    X = np.zeros((len(Qs), s["n"])); Y = np.zeros((len(Qs), s["m"])); I = np.zeros((len(Qs), 8))
    for b in range(len(Qs)):
        r = orc.OracleOSQP().setup(P=s["P"], q=Qs[b], A=s["A"], l=Ls[b], u=Us[b]).solve()
        X[b], Y[b] = r.x, r.y; I[b, 0], I[b, 1], I[b, 2] = r.info.iter, r.info.status_val, r.info.obj_val
    return X, Y, I
X, Y, I = sharded_batch_solve(local, Q, L, U)
np.save(os.path.join(%(out)r, "rank%%d.npy" %% dist.get_rank()), np.concatenate([X, Y, I], axis=1))
dist.destroy_process_group()
'''


def test_sharded_batch_gather_gloo_world2(tmp_path, oracle_mod):
    """N > 1 path on CPU: two gloo ranks each solve their contiguous shard and one
    all_gather returns the whole batch, identical on both ranks and equal to a
    single-process run."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    a = np.load(tmp_path / "rank0.npy"); b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b) and a.shape[0] == 6
    from osqp_amd.problems import mpc_batch
    s, Q, L, U = mpc_batch(6)
    for i in range(6):
        r = oracle_mod.OracleOSQP().setup(P=s["P"], q=Q[i], A=s["A"], l=L[i], u=U[i]).solve()
        assert np.array_equal(a[i, :s["n"]], r.x) and a[i, s["n"] + s["m"]] == r.info.iter


def test_binary_problem_file_round_trip(product_lib, tmp_path):
    """osqp_amd_write_problem / osqp_amd_read_problem (C) and osqp_amd.io (numpy) read each
    other's files; malformed input is rejected with code 3."""
    from osqp_amd import abi, io
    pb, _ = load_golden("primal_infeasibility")
    path = str(tmp_path / "qp.bin")
    io.save_problem(path, **pb)
    L = product_lib
    L.osqp_amd_read_problem.restype = abi.c_int
    L.osqp_amd_read_problem.argtypes = [C.c_char_p, C.POINTER(C.POINTER(abi.OSQPData))]
    L.osqp_amd_write_problem.restype = abi.c_int
    L.osqp_amd_write_problem.argtypes = [C.c_char_p, C.POINTER(abi.OSQPData)]
    L.osqp_amd_free_problem.restype = None
    L.osqp_amd_free_problem.argtypes = [C.POINTER(abi.OSQPData)]
    d = C.POINTER(abi.OSQPData)()
    assert L.osqp_amd_read_problem(path.encode(), C.byref(d)) == 0
    assert (d.contents.n, d.contents.m) == (50, 150)
    path2 = str(tmp_path / "qp2.bin")
    assert L.osqp_amd_write_problem(path2.encode(), d) == 0
    L.osqp_amd_free_problem(d)
    back = io.load_problem(path2)
    from scipy import sparse
    assert np.array_equal(back["q"], pb["q"])
    assert np.array_equal(back["l"], np.clip(pb["l"], -1e30, 1e30))
    assert (abs(back["A"] - sparse.csc_matrix(pb["A"])).nnz == 0)
    assert (abs(back["P"] - sparse.triu(pb["P"], format="csc")).nnz == 0)
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"OSQPAMD1" + b"\x00" * 16)
    assert L.osqp_amd_read_problem(str(bad).encode(), C.byref(d)) == 3


def test_spawn_ranks_sets_env_and_relays_rank0(tmp_path):
    """`bench.py --gpus N` without a launcher starts N ranks itself (osqp_amd/launch.py): every child gets
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, only rank 0's stdout is relayed, the exit code
    is the largest of the children's."""
    import io, json
    from osqp_amd.launch import spawn_ranks
    child = tmp_path / "child.py"
    child.write_text("import os, sys, json\n"
                     "r = int(os.environ['RANK'])\n"
                     "open(os.path.join(sys.argv[1], 'rank%d.json' % r), 'w').write(json.dumps({k: os.environ[k] for k in "
                     "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))\n"
                     "print(json.dumps({'n_gpus': int(os.environ['WORLD_SIZE']), 'rank': r}))\n"
                     "sys.exit(3 if r == 2 and len(sys.argv) > 2 else 0)\n")
    buf = io.StringIO()
    assert spawn_ranks(3, [sys.executable, str(child), str(tmp_path)], stdout=buf) == 0
    assert json.loads(buf.getvalue()) == {"n_gpus": 3, "rank": 0}
    envs = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] == [e["LOCAL_RANK"] for e in envs]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1
    assert spawn_ranks(3, [sys.executable, str(child), str(tmp_path), "fail"], stdout=io.StringIO()) == 3


def test_spawn_ranks_ends_the_siblings_of_a_rank_that_dies(tmp_path):
    """One rank dies early (no device, a failed setup): the others would sit in their next collective until the backend
    times out.  spawn_ranks terminates them and returns the dead rank's code -- within seconds, not minutes."""
    import io, time
    from osqp_amd.launch import spawn_ranks
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\n"
                     "r = int(os.environ['RANK'])\n"
                     "if r == 1: sys.exit(5)\n"
                     "time.sleep(120)\n")
    t0 = time.time()
    assert spawn_ranks(3, [sys.executable, str(child)], stdout=io.StringIO()) == 5
    assert time.time() - t0 < 30


def test_batch_members_above_the_kernel_size_have_no_device_image():
    """BatchOSQP with n > 128 runs one single-QP engine per member: device_arrays() must say so instead of handing a NULL
    handle to the C side (dist.gather_batch_records builds its records from the host results on that path)."""
    from osqp_amd.batch import BatchOSQP
    b = BatchOSQP.__new__(BatchOSQP)
    b._many = [object()]; b._h = None
    with pytest.raises(RuntimeError, match="one single-QP engine per member"):
        b.device_arrays()
    b._many = None


def test_bench_spawns_before_touching_the_gpu():
    """bench.py must start its ranks before torch / the HIP library are imported in the parent (a process that
    initialised the GPU must not fork workers that use it)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("spawn_ranks(") < main.index("import torch") < main.index("import osqp_amd")
    head = src[:src.index("def parse():")]
    assert "import torch" not in head and "import osqp_amd" not in head


def test_every_environment_variable_is_documented():
    """Every OSQP_AMD_* variable the sources read appears in INTEGRATION.md (section F or the diagnostics paragraph)."""
    import glob
    names = set()
    for pat in ("osqp_amd/csrc/*.c", "osqp_amd/csrc/*.hip", "osqp_amd/*.py"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            txt = open(f, errors="ignore").read()
            names |= set(re.findall(r'getenv\("(OSQP_AMD_[A-Z0-9_]+)"\)', txt))
            names |= set(re.findall(r'environ\.get\("(OSQP_AMD_[A-Z0-9_]+)"', txt))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert len(names) > 20 and not missing, missing


def test_debug_dump_helpers_write_the_reference_formats(product_lib, tmp_path, capfd):
    """dump_csc_matrix / dump_vec / print_vec_int (the reference's DDEBUG helpers, src/util.c:393-417, 459-491): 1-based
    "row<TAB>col<TAB>%20.18e" triplets in column order closed by "m<TAB>n<TAB>0", one "%20.18e" per line for vectors."""
    import ctypes as C
    from scipy import sparse
    from osqp_amd import _abi as abi
    M = sparse.csc_matrix(np.array([[4.0, 0.0, 1.5], [0.0, 0.0, -2.0]]))
    h = abi.CscHolder(M)
    L = product_lib
    L.dump_csc_matrix.argtypes = [C.POINTER(abi.csc), C.c_char_p]; L.dump_csc_matrix.restype = None
    L.dump_vec.argtypes = [C.c_void_p, C.c_longlong, C.c_char_p]; L.dump_vec.restype = None
    L.print_vec_int.argtypes = [C.c_void_p, C.c_longlong, C.c_char_p]; L.print_vec_int.restype = None
    f1, f2 = str(tmp_path / "m.txt"), str(tmp_path / "v.txt")
    L.dump_csc_matrix(C.byref(h.struct), f1.encode())
    v = np.array([1.0, -0.5, 1e-300])
    L.dump_vec(v.ctypes.data_as(C.c_void_p), 3, f2.encode())
    lines = open(f1).read().splitlines()
    assert lines == ["1\t1\t%20.18e" % 4.0, "1\t3\t%20.18e" % 1.5, "2\t3\t%20.18e" % -2.0, "2\t3\t%20.18e" % 0.0]
    assert open(f2).read().splitlines() == ["%20.18e" % x for x in v]
    # read back: the triplets are the matrix
    T = np.loadtxt(f1)
    back = sparse.csc_matrix((T[:-1, 2], (T[:-1, 0].astype(int) - 1, T[:-1, 1].astype(int) - 1)), shape=(int(T[-1, 0]), int(T[-1, 1])))
    assert (back != M).nnz == 0
    idx = np.array([3, 1, 2], dtype=np.int64)
    L.print_vec_int(idx.ctypes.data_as(C.c_void_p), 3, b"perm")
