#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc CSVs written by profiles/run_prof.sh into one small JSON per kernel
(mean per launch).  Usage: summarize_pmc.py <gpurun_out/prof_TAG_WL> <out.json>"""
import collections
import csv
import glob
import json
import sys


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(src + "/pmc_*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
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
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
