#!/bin/bash
# usage: scratch/pmc_one.sh <tag> "<counters>" [mb_one args]   -> gpurun_out/<tag>/summary.txt (per-kernel means of each counter)
TAG=$1; CNT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/$TAG/pmc -- python3 $R/scratch/mb_one.py "$@" > $R/gpurun_out/$TAG/out.txt 2> $R/gpurun_out/$TAG/err.txt
F=$(find $R/gpurun_out/$TAG/pmc -name "*counter_collection.csv" | head -1)
python3 - "$F" >> $R/gpurun_out/$TAG/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "conv_" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v[2:]) / max(len(v[2:]), 1)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
rm -rf $R/gpurun_out/$TAG/pmc
