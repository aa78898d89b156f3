"""examples/maxcut_c_abi.c and examples/batch_c_abi.c — the C ABI driven from plain C (no Python, no torch in the process): compiled strictly
against include/sdplr_hip.h; its layout construction and driver logic are dry-run on the CPU against the oracle (which
exports the same interface under the sdplr_oracle_ prefix; the renaming shim exists only inside this test); on a GPU box
the very same source is linked with libsdplr_hip.so and must converge."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "maxcut_c_abi.c")
INC = os.path.join(ROOT, "include")

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")


BATCH_SRC = os.path.join(ROOT, "examples", "batch_c_abi.c")


@pytest.mark.parametrize("src", [SRC, BATCH_SRC])
def test_example_compiles_strictly(tmp_path, src):
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", INC, "-c", src,
                           "-o", str(tmp_path / "ex.o")])


def oracle_shim(tmp_path):
    names = sorted(set(re.findall(r"\bsdplr_hip_[a-z_A-Z0-9]+", open(os.path.join(INC, "sdplr_hip.h")).read())))
    shim = tmp_path / "shim.h"
    shim.write_text("".join(f"#define {n} {n.replace('sdplr_hip_', 'sdplr_oracle_', 1)}\n" for n in names
                            if n not in ("sdplr_hip_h",)))
    return shim


def test_batch_example_logic_on_the_oracle(tmp_path, oracle_abi):
    """The lockstep driver of examples/batch_c_abi.c against the oracle's batch calls (loops over its single-instance
    functions): 9 instances of different sizes, all converge."""
    exe = tmp_path / "batch_oracle"
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-O2", "-include", str(oracle_shim(tmp_path)), "-I", INC, BATCH_SRC, "-L", odir,
                           "-lsdplr_oracle", "-lm", f"-Wl,-rpath,{odir}", "-o", str(exe)])
    out = subprocess.run([str(exe), "9", "6"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK") and "major 8" in out.stdout


@pytest.mark.gpu
def test_batch_example_runs_on_the_device(tmp_path, hip_abi, monkeypatch):
    """The same source on the HIP library: 40 instances side by side (their own route: the resident one)."""
    monkeypatch.delenv("SDPLR_HIP_FORCE_GRAPH", raising=False)
    ldir = os.path.join(ROOT, "sdplrplus.jl_amd", "lib")
    exe = tmp_path / "batch_hip"
    subprocess.check_call(["gcc", "-O2", "-I", INC, BATCH_SRC, "-L", ldir, "-lsdplr_hip", "-lm", f"-Wl,-rpath,{ldir}",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)])
    env = {k: v for k, v in os.environ.items() if k != "SDPLR_HIP_FORCE_GRAPH"}
    out = subprocess.run([str(exe), "40", "6"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK") and "sdplr_hip" in out.stdout.splitlines()[0]


def test_example_logic_on_the_oracle(tmp_path, oracle_abi):
    names = sorted(set(re.findall(r"\bsdplr_hip_[a-z_A-Z0-9]+", open(os.path.join(INC, "sdplr_hip.h")).read())))
    shim = tmp_path / "shim.h"
    shim.write_text("".join(f"#define {n} {n.replace('sdplr_hip_', 'sdplr_oracle_', 1)}\n" for n in names
                            if n not in ("sdplr_hip_h",)))
    exe = tmp_path / "ex_oracle"
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-O2", "-include", str(shim), "-I", INC, SRC, "-L", odir, "-lsdplr_oracle", "-lm",
                           f"-Wl,-rpath,{odir}", "-o", str(exe)])
    out = subprocess.run([str(exe), "200", "6"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK") and "major 8" in out.stdout


@pytest.mark.gpu
def test_example_runs_on_the_device(tmp_path, hip_abi):
    ldir = os.path.join(ROOT, "sdplrplus.jl_amd", "lib")
    exe = tmp_path / "ex_hip"
    subprocess.check_call(["gcc", "-O2", "-I", INC, SRC, "-L", ldir, "-lsdplr_hip", "-lm", f"-Wl,-rpath,{ldir}",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.run([str(exe), "4096", "8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK") and "sdplr_hip" in out.stdout.splitlines()[0]
