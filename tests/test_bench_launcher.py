"""bench.py's own N-rank launch path on CPU: `--gpus 2` with no WORLD_SIZE makes the parent start two fresh ranks
through torch.distributed.run (before it imports torch or the HIP library) and relay rank 0's JSON line; the ranks
run the launcher self-test (gloo + the CPU checker) so that spawn, rendezvous, barrier, max-over-ranks and the gather
are exercised without a GPU.  On the GPU node the same code path runs on the HIP library over RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_launches_two_ranks_and_relays_one_line():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "6", "--warmup", "2", "--selftest-cpu"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["selftest"] is True
    assert len(d["objectives"]) == 2 and d["objectives"][0] != d["objectives"][1]    # two replicas, two graph seeds
    assert d["value"] > 0 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert abs(d["value"] - 2 * 6 / (d["ms_per_step"] * 6 / 1e3)) < 1e-6 * d["value"]   # N·K / max-over-ranks time


def test_world_size_mismatch_is_an_error():
    env = _env()
    env["WORLD_SIZE"] = "4"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-cpu"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr
    assert p.stdout.strip() == ""


def test_parent_does_not_touch_the_gpu_stack():
    """The launcher branch must run before torch / the HIP library are imported (never fork or exec from a process
    that has initialised the GPU): checked on the source — the imports live inside main(), after the branch."""
    src = open(BENCH).read()
    head, _, tail = src.partition("def main():")
    assert "import torch" not in head and "sdplrplus_jl_amd" not in head.replace("import sdplrplus_jl_amd", "")
    assert "import sdplrplus_jl_amd" not in head
    body = tail
    assert body.index("launch_children(args)") < body.index("import sdplrplus_jl_amd")
    assert body.index("launch_children(args)") < body.index("import torch")
