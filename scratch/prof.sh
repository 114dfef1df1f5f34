#!/bin/bash
# usage: scratch/prof.sh <tag> [bench args...]   -> gpurun_out/<tag>_stats.csv (+ pmc)
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-other-models --tune-cache $R/gpurun_out/$TAG/tune.json "$@" > /dev/null 2>&1   # tuning pass outside the profile
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/kt -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-other-models --tune-cache $R/gpurun_out/$TAG/tune.json "$@" > $R/gpurun_out/$TAG/bench_kt.json 2> $R/gpurun_out/$TAG/bench_kt.err || true
find $R/gpurun_out/$TAG/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/$TAG/kernel_stats.csv || true
head -30 $R/gpurun_out/$TAG/kernel_stats.csv | cut -c1-200
