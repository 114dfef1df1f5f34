#!/bin/bash
# usage: scratch/pmc_any.sh <tag> "<counters>" <kernel substring> <script.py> [args]  -> gpurun_out/<tag>/summary.txt
# per-kernel mean of each counter over the launches after the first two
TAG=$1; CNT=$2; PAT=$3; shift; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/$TAG/pmc -- python3 $R/scratch/"$@" > $R/gpurun_out/$TAG/out.txt 2> $R/gpurun_out/$TAG/err.txt
F=$(find $R/gpurun_out/$TAG/pmc -name "*counter_collection.csv" | head -1)
python3 - "$F" "$PAT" >> $R/gpurun_out/$TAG/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:80], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v[2:]) / max(len(v[2:]), 1)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
rm -rf $R/gpurun_out/$TAG/pmc
