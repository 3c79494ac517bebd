#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc CSVs written by profiles/run_prof.sh into one small JSON per kernel
(mean per launch).  Usage: summarize_pmc.py <gpurun_out/prof_TAG_WL> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_source_hash  # noqa: E402  (the bench refuses summaries of other kernel sources)


def main(src, dst):
    # the bench also launches the same kernels on small frames (multi-frame driver leg): keep, per
    # kernel, only the launches with the largest grid = the full frames of the named workload
    rows = []
    for f in glob.glob(src + "/pmc_*/**/*_counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    if not rows:  # never replace a committed summary with an empty one (the GPU run may not have happened)
        raise SystemExit(f"no counter files under {src}: nothing written")
    biggest = collections.defaultdict(int)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        biggest[k] = max(biggest[k], int(r["Grid_Size"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if True:
            k = r["Kernel_Name"].split("(")[0]
            if int(r["Grid_Size"]) != biggest[k]:
                continue
            agg[k]["_GRID"] = [float(r["Grid_Size"])]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_VGPR"] = [float(r["VGPR_Count"])]
            agg[k]["_SGPR"] = [float(r["SGPR_Count"])]
            agg[k]["_LDS"] = [float(r["LDS_Block_Size"])]
    out = {k: {c: sum(v) / len(v) for c, v in sorted(d.items())} for k, d in agg.items() if k.startswith(("rtk", "void rtk"))}
    for k, d in out.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            # MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
            # half the bytes of wide (16 B/lane) coalesced reads -> doubled (upper estimate for this kernel,
            # whose dominant HBM stream is the 16 B/lane jitter read).
            d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
            d["hbm_read_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * 2
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
            d["hbm_bytes_per_launch"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
    # top-level figure bench.py reports as roofline.traffic: the dominant kernel's full-frame launch
    # (the counter-free instantiation of the packet kernel: <false, true> is the 64-spp one, <false, false> the rest)
    names = [k for k in out if k.startswith("void rtk::render_packet_kernel<false")] or \
            [k for k in out if k.startswith("void rtk::render_kernel<")]
    name = max(names, key=lambda k: out[k].get("SQ_INSTS_VALU", 0.0)) if names else ""
    dom = out.get(name, {})
    out["dominant_kernel"] = name
    out["hbm_bytes_per_launch"] = dom.get("hbm_bytes_per_launch")
    out["kernel_source_hash"] = kernel_source_hash()
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
