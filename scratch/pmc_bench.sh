#!/bin/bash
# usage: scratch/pmc_bench.sh <tag> "<counters>" <kernel substring>
TAG=$1; CNT=$2; PAT=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/$TAG/pmc -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-graph --no-autotune > $R/gpurun_out/$TAG/out.txt 2> $R/gpurun_out/$TAG/err.txt
F=$(find $R/gpurun_out/$TAG/pmc -name "*counter_collection.csv" | head -1)
python3 - "$F" "$PAT" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if sys.argv[2] not in k: continue
    print(k, {c: round(sum(v)/len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
