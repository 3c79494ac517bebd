#!/usr/bin/env python3
"""Pack the raw arrays dumped by oracle/_ref/eigen_probe into tests/golden/eigen_probe.npz.

Test infrastructure only.  The probe (oracle/eigen_probe.cpp) is compiled against the
reference's own nanobind-free headers and vendored Eigen, from the read-only mount; this
script only reshapes its output into a compressed fixture that travels with the repo.
Usage: make_golden.py <probe_out_dir> <out.npz>
"""
import os
import sys

import numpy as np

DT = {"f64": np.float64, "i32": np.int32, "u32": np.uint32}


def main(src, dst):
    arrays = {}
    for name in sorted(os.listdir(src)):
        stem, ext = name.rsplit(".", 1)
        arrays[stem] = np.fromfile(os.path.join(src, name), dtype=DT[ext])
    np.savez_compressed(dst, **arrays)
    print(f"wrote {dst}: {len(arrays)} arrays, {os.path.getsize(dst)} bytes")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
