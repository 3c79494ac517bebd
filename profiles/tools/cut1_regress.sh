#!/bin/bash
# cut_mode 1 (one cut level for frames of 2 M .. 12 M rays) against no cuts: one mesh of growing size at 400x225x50
# through the frame driver (warm call), and the 1 M-triangle plate at 2048^2 with 1 and 2 spp.
cd "${GRAFT_REPO_ROOT:-.}"
for n in 16 32 45 64 90 128 256; do
  a=$(timeout -k 10 100 python3 profiles/tools/cold_job.py 60 $n | tail -1) || exit 1
  b=$(RT_CUT1_MIN_RAYS=1e12 timeout -k 10 100 python3 profiles/tools/cold_job.py 60 $n | tail -1) || exit 1
  echo "one level: $a"
  echo "no cuts  : $b"
done
