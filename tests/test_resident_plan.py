"""Host side of the resident PCG set-up (engine.hip: build_resident) without a device: the pattern of
K = P + sigma I + A' rho A against scipy, the row partition, the line-padded positions in the exchanged vector and the
register layout (every entry of K in exactly one slot of a thread of its rows' workgroup)."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy import sparse

import osqp_amd
from osqp_amd import _abi as abi


def _plan(P, A, nwg=256):
    L = osqp_amd.lib()
    f = L.hipeng_resident_plan
    f.restype = C.c_int
    f.argtypes = [C.POINTER(abi.csc), C.POINTER(abi.csc), C.c_int, C.POINTER(C.c_longlong), C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    Pu = abi.CscHolder(sparse.triu(P, format="csc")); Ah = abi.CscHolder(A)
    n = P.shape[0]
    stats = (C.c_longlong * 8)()
    assert f(C.byref(Pu.struct), C.byref(Ah.struct), nwg, stats, None, None, None, 0, None, None, None) == 0
    st = list(stats)
    if not st[0]:
        return st, None
    nnz, E = st[3], st[1]
    Kptr = np.zeros(n + 1, dtype=np.int32); Kcol = np.zeros(nnz, dtype=np.int32); kdst = np.zeros(nnz, dtype=np.int32)
    cap = abs(nwg)                      # (nwg < 0: the engine sizes the grid itself, at most -nwg workgroups)
    rowpos = np.zeros(n, dtype=np.uint16); slotcol = np.zeros(cap * E * st[7], dtype=np.uint16); wg4 = np.zeros(4 * cap, dtype=np.int32)
    assert f(C.byref(Pu.struct), C.byref(Ah.struct), nwg, stats, Kptr.ctypes.data, Kcol.ctypes.data, kdst.ctypes.data, nnz,
             rowpos.ctypes.data, slotcol.ctypes.data, wg4.ctypes.data) == 0
    return st, dict(Kptr=Kptr, Kcol=Kcol, kdst=kdst, rowpos=rowpos, slotcol=slotcol, wg=wg4.reshape(cap, 4))


def _qp(n, m, seed, dens):
    rng = np.random.default_rng(seed)
    A = sparse.random(m, n, density=dens, random_state=seed, data_rvs=rng.standard_normal, format="csc")
    B = sparse.random(n, n, density=dens, random_state=seed + 1, data_rvs=rng.standard_normal, format="csc")
    P = (B @ B.T + 0.05 * sparse.eye(n)).tocsc()
    return P, A


@pytest.mark.parametrize("n,m,dens,nwg", [(700, 1100, 0.01, 256), (1500, 400, 0.004, 256), (300, 200, 0.05, 64), (2, 3, 0.9, 256)])
def test_plan_pattern_partition_and_layout(n, m, dens, nwg):
    os.environ["OSQP_AMD_RESIDENT_MIN_N"] = "1"
    try:
        P, A = _qp(n, m, 7 + n, dens)
        st, pl = _plan(P, A, nwg)
    finally:
        os.environ.pop("OSQP_AMD_RESIDENT_MIN_N")
    assert st[0] == 1, st
    E, npad, nnz, PT = st[1], st[2], st[3], st[7]
    # pattern of K: P (both triangles), the diagonal, A'A
    Pf = sparse.triu(P) + sparse.triu(P, 1).T
    pat = ((abs(Pf) + sparse.eye(n) + abs(A).T @ abs(A)) != 0).tocsr()
    pat.sort_indices()
    assert nnz == pat.nnz
    assert np.array_equal(pl["Kptr"], pat.indptr) and np.array_equal(pl["Kcol"], pat.indices)
    # partition: contiguous blocks of rows covering 0..n, at most 61 rows and 448 E entries, positions in lines of their own
    wg = pl["wg"]
    row = 0; pos = 0
    for r0, nr, cnt, p0 in wg:
        if nr == 0:
            assert cnt == 0
        else:
            assert r0 == row and 0 < nr <= 61 and cnt == pl["Kptr"][r0 + nr] - pl["Kptr"][r0] and cnt <= PT * E
            assert np.array_equal(pl["rowpos"][r0:r0 + nr], p0 + np.arange(nr))
            row += nr
        assert p0 == pos and p0 % 16 == 0            # a 128-byte line is written by one workgroup only
        pos += (nr + 3 + 15) // 16 * 16              # rows + three riding partials, padded to whole lines
    assert row == n and pos == npad and npad <= 16384
    assert st[4] == int((wg[:, 1] > 0).sum()) and st[5] == wg[:, 1].max() and st[6] == wg[:, 2].max()
    # register layout: every entry of K in exactly one slot, the slot belongs to a thread of the owner of its row, and
    # holds the position of the entry's column
    kdst = pl["kdst"].astype(np.int64)
    assert len(np.unique(kdst)) == nnz
    rows = np.repeat(np.arange(n), np.diff(pl["Kptr"]))
    owner = np.zeros(n, dtype=np.int64)
    for g, (r0, nr, cnt, p0) in enumerate(wg):
        owner[r0:r0 + nr] = g
    g_of_slot = kdst // (E * PT)
    k_of_slot = (kdst // PT) % E
    t_of_slot = kdst % PT
    assert np.array_equal(g_of_slot, owner[rows])
    assert ((t_of_slot * E + k_of_slot) < wg[g_of_slot, 2]).all()           # inside the workgroup's list of entries
    assert np.array_equal(pl["slotcol"][kdst], pl["rowpos"][pl["Kcol"]])
    # a thread's entries are E consecutive entries of its workgroup's row-major list (in some order)
    le = (t_of_slot * E + k_of_slot)
    base = pl["Kptr"][wg[g_of_slot, 0]]
    entry_index = np.arange(nnz) - base
    assert np.array_equal(entry_index // E, le // E)


def test_plan_refuses_what_does_not_fit():
    # n beyond the row capacity of the machine: 64 CUs x 61 rows
    P, A = _qp(4000, 100, 3, 0.001)
    st, _ = _plan(P, A, 64)
    assert st[0] == 0
    # a row of A with 8192 or more entries (its outer product alone overflows the register files)
    n = 9000
    A = sparse.vstack([sparse.csc_matrix(np.ones((1, n))), sparse.eye(n, format="csc")], format="csc")
    P = sparse.eye(n, format="csc")
    st, _ = _plan(P, A, 256)
    assert st[0] == 0
    # below the default size threshold
    P, A = _qp(100, 50, 5, 0.1)
    st, _ = _plan(P, A, 256)
    assert st[0] == 0


def test_the_grid_is_sized_to_the_problem():
    """nwg < 0: the grid the engine itself chooses on a machine of -nwg CUs.  Every exchange inside a resident launch is an
    all-to-all between the participating CUs, so a small K gets a small grid: the fewest workgroups that keep 61 rows and
    16 x 448 entries each (8 x 448 where halving the entries would not halve the grid); only a K that does not fit 256 such
    blocks lets the entries per thread grow."""
    grids = []
    for n, m, dens in ((300, 200, 0.02), (900, 700, 0.02), (2500, 1500, 0.004)):
        P, A = _qp(n, m, n, dens)
        st, pl = _plan(P, A, -256)
        assert st[0] == 1
        nwg = st[4]                      # workgroups that own rows = the grid
        assert pl["wg"][:nwg, 1].min() >= 1 and (pl["wg"][nwg:, 1] == 0).all()
        assert st[1] <= 16 and st[5] <= 61 and st[6] <= st[1] * 448
        assert nwg >= -(-n // 61)
        # one workgroup fewer would not do at E = 16
        if nwg > -(-n // 61):
            assert st[3] > (nwg - 1) * 16 * 448 * 0.5     # (greedy contiguous packing wastes at most a row per block)
        assert st[2] == sum(((int(r) + 3 + 15) // 16) * 16 for r in pl["wg"][:nwg, 1])   # the exchanged vector shrinks with the grid
        grids.append(nwg)
    assert grids[0] < 16 and grids[0] < grids[1] < 256
    # a fixed grid of 256 for the same problem: same K, more (mostly empty) participants
    P, A = _qp(300, 200, 300, 0.02)
    st256, _ = _plan(P, A, 256)
    assert st256[3] == _plan(P, A, -256)[0][3] and st256[2] > _plan(P, A, -256)[0][2]


def test_the_plan_does_not_depend_on_the_number_of_setup_threads(monkeypatch):
    """build_resident splits the symbolic K by rows and the register layout by workgroups over host threads
    (OSQP_AMD_SETUP_THREADS): every array of the plan is the same for 1, 3 and 8 of them."""
    P, A = _qp(3000, 2000, 11, 0.003)
    plans = []
    for k in ("1", "3", "8"):
        monkeypatch.setenv("OSQP_AMD_SETUP_THREADS", k)
        st, pl = _plan(P, A, -256)
        assert st[0] == 1
        plans.append((st, pl))
    for st, pl in plans[1:]:
        assert st == plans[0][0]
        for key in pl:
            assert np.array_equal(pl[key], plans[0][1][key]), key
