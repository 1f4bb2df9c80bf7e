#!/usr/bin/env python3
"""Capture the reference's own test fixtures as small JSON golden files.

The reference (EmreAdabag/osqp, /root/reference) keeps its unit-test data as
Python *generators* (tests/<name>/generate_problem.py) that hand numpy/scipy
objects to tests/utils/codegen_utils.py, which writes a git-ignored C header.
This script imports those generators IN THE BUILD CONTAINER ONLY, intercepts the
two codegen entry points (generate_problem_data / generate_data,
tests/utils/codegen_utils.py:172,351) and dumps what they were given -- inputs
and the hard-coded expected solutions -- to tests/golden/<name>.json.

Nothing is written under /root/reference, and no reference source text is
stored: only the numeric data the generators define.  Re-run with
    python tests/golden/make_golden.py
(requires /root/reference; the committed JSON files are what travels).

`primal_infeasibility` uses scipy.randn, which SciPy removed; the generator is
given numpy.random.randn under that name (q, l, u there are unseeded upstream,
so this is a one-off capture by construction).
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import scipy
from scipy import sparse

REF_TESTS = "/root/reference/tests"
OUT = os.path.dirname(os.path.abspath(__file__))

NAMES = ["basic_qp", "basic_qp2", "lin_alg", "non_cvx",
         "primal_dual_infeasibility", "primal_infeasibility", "solve_linsys",
         "unconstrained", "update_matrices"]


def enc(v):
    """JSON encoding; +-inf are kept as the strings 'inf'/'-inf'."""
    if sparse.issparse(v):
        c = sparse.csc_matrix(v)
        c.sort_indices()
        return {"__csc__": True, "m": int(c.shape[0]), "n": int(c.shape[1]),
                "indptr": [int(t) for t in c.indptr],
                "indices": [int(t) for t in c.indices],
                "data": [enc(float(t)) for t in c.data]}
    if isinstance(v, np.ndarray):
        return {"__vec__": True, "shape": list(v.shape),
                "data": [enc(float(t)) for t in v.ravel()]}
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if np.isinf(v):
            return "inf" if v > 0 else "-inf"
        return v
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, str):
        return v
    raise TypeError(type(v))


captured = {}
current = [None]


def cap_problem(P, q, A, l, u, problem_name, sols_data={}):
    captured[current[0]] = {
        "kind": "problem", "P": enc(sparse.triu(P, format="csc")), "q": enc(np.asarray(q, float)),
        "A": enc(A), "l": enc(np.asarray(l, float)), "u": enc(np.asarray(u, float)),
        "sols": {k: enc(v) for k, v in sols_data.items()}}


def cap_data(problem_name, sols_data):
    captured[current[0]] = {"kind": "data",
                            "data": {k: enc(v) for k, v in sols_data.items()}}


def main():
    sys.path.insert(0, REF_TESTS)
    sys.dont_write_bytecode = True
    cu = importlib.import_module("utils.codegen_utils")
    cu.generate_problem_data = cap_problem
    cu.generate_data = cap_data
    if not hasattr(scipy, "randn"):
        scipy.randn = np.random.randn
    for name in NAMES:
        current[0] = name
        importlib.import_module(name + ".generate_problem")
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(captured[name], f, indent=0, separators=(",", ":"))
        print("captured", name, captured[name]["kind"])


if __name__ == "__main__":
    main()
