#!/bin/bash
# usage: scratch/pmc_bench.sh <tag> "<counters>" <kernel substring>
# PMC pass over the default bench workload (eager launches, launch shapes from a tuning pass done outside the profile)
TAG=$1; CNT=$2; PAT=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-3d --no-other-models --tune-cache $R/gpurun_out/$TAG/tune.json > /dev/null 2>&1
timeout -k 10 400 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/$TAG/pmc -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-3d --no-other-models --no-graph --tune-cache $R/gpurun_out/$TAG/tune.json > $R/gpurun_out/$TAG/out.txt 2> $R/gpurun_out/$TAG/err.txt
F=$(find $R/gpurun_out/$TAG/pmc -name "*counter_collection.csv" | head -1)
python3 - "$F" "$PAT" $R/gpurun_out/$TAG/summary.json <<'PY'
import csv, sys, collections, json, hashlib, os
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    if sys.argv[2] not in k: continue
    out[k] = {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in d.items()}
    print(k, {c: round(sum(v)/len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
csrc = os.path.join(os.environ["GRAFT_REPO_ROOT"], "self-supervised-medical-image-registration-using-deep-optical-flow-estimation-with-brain-mri-data_amd", "csrc")
sha = hashlib.sha1(b"".join(open(os.path.join(csrc, f), "rb").read() for f in sorted(os.listdir(csrc)) if (f.startswith("conv_") and f.endswith(".hip")) or f == "mireg_common.h")).hexdigest()[:12]
json.dump({"csrc_sha1": sha, "command": "scratch/pmc_bench.sh (rocprofv3 --pmc over bench.py --no-graph, eager launches)", "kernels": out}, open(sys.argv[3], "w"), indent=1)
PY
