"""Host-side pieces around the path: the batch generator (exps/gen_batch_test.jl) and the rounding callbacks of the
experiment runner (exps/test.jl:67-105)."""
import json
import os
import sys

import numpy as np

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import problems, rounding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_gen_batch_test_writes_the_reference_batch(tmp_path):
    import gen_batch_test
    out = tmp_path / "batch_test.txt"
    gen_batch_test.main(["--out", str(out)])
    lines = out.read_text().strip().splitlines()
    assert len(lines) == 9                                           # exps/batch_test.txt:1-9
    for i, line in enumerate(lines, 1):
        assert line.startswith("ulimit -d 16777216;")
        assert f'--seed 0 --graph "G{i}" --problem "MaxCut" --ptol 0.01 --objtol 0.01 --rank 10' in line
    man = json.loads((tmp_path / "batch_test.json").read_text())
    assert [e["graph"] for e in man] == [f"G{i}" for i in range(1, 10)] and man[0]["rank"] == 10
    gen_batch_test.main(["--out", str(out), "--extra-gnp", "55"])
    man = json.loads((tmp_path / "batch_test.json").read_text())
    assert len(man) == 64 and man[9]["graph"] == "gnp:800:0.06:10" and man[-1]["graph"] == "gnp:800:0.06:64"


def test_rounding_callbacks():
    rng = np.random.Generator(np.random.PCG64(3))
    # two cliques joined by one edge: the planted cut / bisection is known
    n = 12
    A = np.zeros((n, n))
    A[:6, :6] = 1; A[6:, 6:] = 1
    np.fill_diagonal(A, 0)
    A[5, 6] = A[6, 5] = 1
    import scipy.sparse as sp
    A = sp.csc_matrix(A)
    L = problems._laplacian(A, 1.0)
    x = np.array([1.0] * 6 + [-1.0] * 6)
    assert rounding.eval_cut(L, x) == 1.0                            # one edge crosses
    R = np.outer(x, [1.0, 0.0]) + 1e-3 * rng.standard_normal((n, 2)) # a rank-one bisection solution
    assert rounding.minimumbisection_rounding(A, R, rng) == 1.0
    # MaxCut of a bipartite graph = all edges; the exact solution R = ±1 vector rounds to it
    B = sp.csc_matrix(np.kron(np.array([[0, 1], [1, 0]]), np.ones((4, 4))))
    xb = np.array([1.0] * 4 + [-1.0] * 4)
    assert rounding.maxcut_rounding(B, np.outer(xb, [0.6, 0.8]), rng) == 16.0
    # on a solved instance the rounded cut lies between half the SDP bound and the bound
    G = problems.gnp_graph(60, 0.2, 1)
    from oracle import oracle
    res = sj.sdplr(data=problems.maxcut_data(G), r=6, abi=oracle.abi(), printlevel=0, seed=0, prior_trace_bound=60.0)
    cut = rounding.maxcut_rounding(G, res["Rt"], rng)
    assert 0.878 * 0.95 * (-res["obj"]) <= cut <= -res["max_dual_value"] * (1 + 1e-2)
