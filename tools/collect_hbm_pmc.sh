#!/bin/bash
# HBM-side traffic per kernel of a workload given as "python3 <script> <args>": two rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE; separate runs, kernel trace only next to them) -> gpurun_out/<tag>/pmc_hbm.json (tools/pmc_summary.py)
#   tools/collect_hbm_pmc.sh <tag> <script> [args...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 "$@" > $OUT/f.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 "$@" > $OUT/w.log 2>&1 || echo "write pass failed"
python3 tools/pmc_summary.py $(ls $OUT/f/*/*counter_collection.csv | head -1) $(ls $OUT/w/*/*counter_collection.csv | head -1) > $OUT/pmc_hbm.json
rm -rf $OUT/f $OUT/w
python3 - $OUT/pmc_hbm.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
rows = []
for k, v in d.items():
    if not k.startswith("k_"):
        continue
    f = 2 * v.get("FETCH_SIZE", {}).get("mean_KiB_per_launch", 0) / 1024
    w = v.get("WRITE_SIZE", {}).get("mean_KiB_per_launch", 0) / 1024
    n = v.get("FETCH_SIZE", {}).get("launches", 0)
    rows.append((f + w, k, f, w, n))
for t, k, f, w, n in sorted(rows, reverse=True):
    print("%-28s fetch %7.2f MB  write %7.2f MB per launch  (%d launches)" % (k[:28], f, w, n))
PY
