#!/usr/bin/env python
"""profiles/<round>_hbm_traffic.json from the two PMC summaries (tools/rocprof_summary.py pmc): memory-side bytes of the
dominant kernel (all gemm_f64_kernel instantiations), per launch.  FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE counts
half of the bytes of 16-byte-per-lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section; confirmed here on
symv_partial_kernel, which reads an n x np matrix exactly once), hence the factor 2."""
import glob
import hashlib
import json
import os
import re
import sys


def csrc_sha16():
    """hash of the device sources the counters were collected with: bench.py quotes the record only while it still matches"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynadjust_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def parse(path, counter):
    tot, n, cur = 0.0, 0, None
    for line in open(path):
        if not line.startswith(" ") and not line.startswith("#"):
            cur = line.strip()
        m = re.match(r"\s+dispatches (\d+)", line)
        if m and cur and "gemm_f64_" in cur:
            n += int(m.group(1))
        m = re.match(r"\s+%s\s+sum ([0-9.e+-]+)" % counter, line)
        if m and cur and "gemm_f64_" in cur:
            tot += float(m.group(1))
    return tot, n


if __name__ == "__main__":
    fetch_txt, write_txt, out, workload = sys.argv[1:5]
    f, nf = parse(fetch_txt, "FETCH_SIZE")
    w, nw = parse(write_txt, "WRITE_SIZE")
    assert nf == nw and nf > 0
    rec = {"workload": workload, "kernel": "gemm_f64_dma_kernel + gemm_f64_kernel (all instantiations of the tile GEMM)", "launches": nf,
           "fetch_size_kib_sum": f, "write_size_kib_sum": w, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 / nf,
           "source": [fetch_txt, write_txt], "csrc_sha16": csrc_sha16()}
    json.dump(rec, open(out, "w"), indent=1)
    print(rec)
