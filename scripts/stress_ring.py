"""dev tool: randomized sequences of library calls on two handles of the same instance — one keeping the L-BFGS history in
ring form inside its loops (the default), one storing it (SDPLR_HIP_NO_RING=1, read at construction) — compared BITWISE
after every call (R, the call's return values) and in full at the end (R, G, dirt, every s_j / y_j, ρ, y, λ).  The
calls are whatever a driver may interleave with inner loops: more loops (short, long, leaving by the relative-decrease
test, with a tolerance that stops them at once), lbfgs_clear!, fg!, g!, the dual bound, the λ update, a major iteration in
one call, looks at and writes to the arena (G, a history slot, λ), the stand-alone lbfgs_dir! / lbfgs_update!, a line
search, a rank reset.  Exercises the hand-over between ring form and stored form (csrc/sdplr_hip.hip: ring_keep,
ensure_canonical, ring_drop_for_clear) far beyond what tests/test_gpu_ring.py enumerates.
    SDPLR_HIP_FORCE_GRAPH=1 python scripts/stress_ring.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("SDPLR_HIP_FORCE_GRAPH", "1")
import numpy as np
import sdplrplus_jl_amd as sj
from sdplrplus_jl_amd import cabi, problems
from helpers import make_solver

BUILD = {"maxcut": problems.maxcut_data, "minbis": problems.minimum_bisection_data, "cutnorm": problems.cutnorm_data}


def run(seed, CASES, hip=None, nmax=900):
    """→ (mismatching cases, sequences cut short by a non-finite state, loops on the ring form, rings turned back)"""
    hip = hip or sj.load_hip()
    rng = np.random.Generator(np.random.PCG64(seed))
    bad = 0
    ring_loops = mats = degenerate = 0
    for case in range(CASES):
        fam = list(BUILD)[int(rng.integers(3))]
        n = int(rng.integers(60, nmax)); r = int(rng.choice([4, 6, 8, 10, 16, 32, 64])); h = int(rng.integers(1, 5))
        g = problems.gnp_graph(n, float(rng.uniform(4.0, 12.0)) / n, int(rng.integers(1 << 30)))
        if g.nnz == 0:
            continue
        data = BUILD[fam](g)
        normC, normb = data.normC(), float(np.linalg.norm(data.b))
        os.environ.pop("SDPLR_HIP_NO_RING", None)
        A, _ = make_solver(hip, data, r, seed=case, h=h)
        os.environ["SDPLR_HIP_NO_RING"] = "1"
        B, _ = make_solver(hip, data, r, seed=case, h=h)
        os.environ.pop("SDPLR_HIP_NO_RING", None)
        st = [A.fg(normC, normb), B.fg(normC, normb)]
        assert st[0] == st[1]
        st = list(st[0])
        log = []
        ok = True
        v0 = rng.standard_normal(data.n)
        for step in range(int(rng.integers(6, 16))):
            op = str(rng.choice(["loop", "loop", "loop", "loop_reldelta", "loop_gtol", "clear", "fg", "g", "dual", "update_lambda",
                                 "major", "look", "write_G", "write_S", "write_lam", "dir_update", "linesearch", "reset_rank"]))
            k = int(rng.integers(1, 14))
            arg = float(rng.uniform(0.5, 2.0))
            res = []
            for s_ in (A, B):
                if op == "loop":
                    out = s_.inner_loop(normC, normb, True, True, False, 0.0, -1e300, k, 0.0, *st)
                elif op == "loop_reldelta":
                    out = s_.inner_loop(normC, normb, True, True, False, 0.0, 1e300, k, 0.0, *st)
                elif op == "loop_gtol":
                    out = s_.inner_loop(normC, normb, True, True, False, 1e30, -1e300, k, 0.0, *st)
                elif op == "clear":
                    s_.lbfgs_clear(); out = ()
                elif op == "fg":
                    out = s_.fg(normC, normb)
                elif op == "g":
                    s_.g(); out = tuple(s_.norms(normC, normb, True, True))
                elif op == "dual":
                    out = s_.dual_obj(float(data.n), 0, v0)
                elif op == "update_lambda":
                    s_.update_lambda(); out = ()
                elif op == "major":
                    out = s_.major_iteration(normC, normb, True, True, False, bool(k & 1), 2.0 * arg, 0.0, -1e300, k, 0.0)
                elif op == "look":
                    j = k % h
                    out = (float(np.sum(s_.get_factor(cabi.F_LBFGS_S + j))), float(np.sum(s_.get_factor(cabi.F_LBFGS_Y + j))),
                           float(np.sum(s_.Gt)), float(np.sum(s_.get_factor(cabi.F_DIRT))))
                elif op == "write_G":
                    s_.set_factor(cabi.F_GT, arg * s_.Gt); out = ()
                elif op == "write_S":
                    j = k % h
                    s_.set_factor(cabi.F_LBFGS_S + j, arg * s_.get_factor(cabi.F_LBFGS_S + j)); out = ()
                elif op == "write_lam":
                    s_.λ = s_.λ + 0.01 * arg; out = ()
                elif op == "dir_update":
                    d = s_.lbfgs_dir(True)
                    s_.lbfgs_update(1e-3 * arg)
                    out = (d,)
                elif op == "linesearch":
                    d = s_.lbfgs_dir(True)
                    out = (d,) + tuple(s_.linesearch(1.0)) if d < 0 else (d,)
                elif op == "reset_rank":
                    s_.reset_rank(r)
                    s_.set_factor(cabi.F_RT, np.full((data.n, r), 0.3) + 0.01 * np.arange(r)[None, :])
                    out = ()
                res.append((out, s_.Rt.copy()))
            log.append((op, k))
            o0, o1 = np.asarray(res[0][0], dtype=float), np.asarray(res[1][0], dtype=float)
            same = o0.shape == o1.shape and np.array_equal(o0, o1, equal_nan=True) and np.array_equal(res[0][1], res[1][1], equal_nan=True)
            if not same:
                ok = False
                print(f"MISMATCH case {case} {fam} n={n} r={r} h={h} after {log}: {res[0][0]} vs {res[1][0]}", flush=True)
                break
            if not (np.all(np.isfinite(o0)) and np.all(np.isfinite(res[0][1]))):
                degenerate += 1          # (the random calls drove the state to NaN / inf — identically on both handles: next case)
                break
            if op in ("loop", "loop_reldelta", "loop_gtol", "major") and len(res[0][0]) >= 3:
                st = list(res[0][0][:3])
            elif op in ("fg",):
                st = list(res[0][0])
            elif op in ("g", "write_lam", "update_lambda", "write_G", "reset_rank", "dir_update", "linesearch"):
                st = [list(s_.fg(normC, normb)) for s_ in (A, B)]
                if not np.array_equal(np.asarray(st[0]), np.asarray(st[1]), equal_nan=True):
                    ok = False
                    print(f"MISMATCH (fg after {op}) case {case} {fam} n={n} r={r} h={h} after {log}", flush=True)
                    break
                st = st[0]
        if ok:
            for name, get in (("G", lambda s_: s_.Gt), ("D", lambda s_: s_.get_factor(cabi.F_DIRT)), ("y", lambda s_: s_.y),
                              ("lam", lambda s_: s_.λ), ("rho", lambda s_: s_.get_vec(cabi.V_LBFGS_RHO))):
                if not np.array_equal(get(A), get(B), equal_nan=True):
                    ok = False
                    print(f"MISMATCH in {name} at the end: case {case} {fam} n={n} r={r} h={h} after {log}", flush=True)
            for j in range(h):
                for slot in (cabi.F_LBFGS_S, cabi.F_LBFGS_Y):
                    if not np.array_equal(A.get_factor(slot + j), B.get_factor(slot + j), equal_nan=True):
                        ok = False
                        print(f"MISMATCH in history slot {slot + j} at the end: case {case} {fam} n={n} r={r} h={h} after {log}", flush=True)
        sa = A.stats()
        ring_loops += sa["ring_history_loops"]; mats += sa["ring_materializations"]
        assert B.stats()["ring_history_loops"] == 0
        bad += 0 if ok else 1
        A.close(); B.close()
    return bad, degenerate, ring_loops, mats


if __name__ == "__main__":
    CASES = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    bad, degenerate, ring_loops, mats = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, CASES)
    print(f"cases {CASES}: mismatching {bad}; sequences cut short by a non-finite state {degenerate}; loops on the ring form {ring_loops}, rings turned back {mats}")
    sys.exit(1 if bad else 0)
