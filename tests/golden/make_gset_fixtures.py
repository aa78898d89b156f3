"""Re-encodes the reference's Gset fixture graphs (exps/data/MaxCut/G1..G9.mat, MATLAB v7.3 / HDF5,
group /A with CSC arrays data/ir/jc) as one compact edge-list archive, tests/golden/gset_G1_G9.npz.

Data only (input graphs; the reference stores no expected outputs for them).  Needs /root/reference and
/opt/conda/bin/h5dump, so it runs in the build container, not on the GPU box; the .npz is committed.
"""
import os
import subprocess
import tempfile

import numpy as np

SRC = "/root/reference/exps/data/MaxCut"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gset_G1_G9.npz")


def dump(path, dset, dtype):
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        subprocess.check_call(["/opt/conda/bin/h5dump", "-d", dset, "-b", "LE", "-o", f.name, path],
                              stdout=subprocess.DEVNULL)
        return np.fromfile(f.name, dtype=dtype)


def main():
    out = {}
    for k in range(1, 10):
        p = os.path.join(SRC, f"G{k}.mat")
        data, ir, jc = dump(p, "/A/data", np.float64), dump(p, "/A/ir", np.uint64), dump(p, "/A/jc", np.uint64)
        n = jc.size - 1
        cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(jc.astype(np.int64)))
        rows = ir.astype(np.int64)
        keep = rows < cols                      # symmetric, zero diagonal: keep the upper triangle
        assert 2 * keep.sum() == rows.size and np.all(data == 1.0)
        out[f"G{k}"] = np.stack([rows[keep], cols[keep]], axis=1).astype(np.uint16)
        out[f"G{k}_n"] = np.int64(n)
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
