"""On-disk formats (exps/data_utils.jl): SDPA round trip, SDPLR layout, initial-solution file, edge lists."""
import numpy as np
import scipy.sparse as sp

import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import io, problems


def test_sdpa_round_trip(tmp_path):
    A = problems.gnp_graph(12, 0.4, 1)
    C, As, bs = problems.maxcut(A)
    p = str(tmp_path / "g.dat-s")
    io.write_problem_sdpa(p, C, As, bs)
    C2, As2, bs2 = io.read_sdpa(p)
    assert np.allclose(bs, bs2) and abs(C - C2).max() < 1e-15
    for a, b in zip(As, As2):
        assert np.allclose(a.toarray(), b.toarray())
    head = open(p).read().split("\n")[:3]
    assert head == ["12", "1", "12"]                   # m, number of blocks, block size (data_utils.jl:35-37)


def test_sdplr_and_initial_solution_files(tmp_path):
    A = problems.gnp_graph(6, 0.6, 2)
    C, As, bs = problems.minimum_bisection(A)          # sparse cost, COO rows, one low-rank row
    p = str(tmp_path / "g.sdplr")
    io.write_problem_sdplr(p, C, As, bs)
    lines = open(p).read().split("\n")
    assert lines[:3] == ["7", "1", "6"] and lines[4] == "1"
    assert lines[5].startswith("0 1 s ")               # the cost block, sparse (data_utils.jl:57)
    assert any(l == "7 1 l 1" for l in lines)          # the rank-1 constraint as a low-rank block (:69)
    R = np.arange(12.0).reshape(6, 2)
    q = str(tmp_path / "init")
    io.write_initial_solution(q, R, np.array([0.5, -1.0]))
    t = open(q).read()
    assert t.startswith("dual variable 2\n0.5\n-1.0\nprimal variable 1 s 6 2 2\n0.0\n2.0\n")   # column-major R
    assert "special sigma " in t and t.endswith("special scale 1.0\n")


def test_edge_list_reader(tmp_path):
    p = tmp_path / "g.txt"
    p.write_text("# comment\n1 2\n2 3 2.5\n3 3\n")
    A = io.read_edge_list(str(p))
    assert A.shape == (3, 3) and (A - A.T).nnz == 0 and A[1, 2] == 2.5 and A[2, 2] == 0
    g = tmp_path / "G.txt"
    g.write_text("4 2\n1 2 1\n3 4 1\n")
    B = io.read_edge_list(str(g), gset_header=True)
    assert B.shape == (4, 4) and B.nnz == 4
