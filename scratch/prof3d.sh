#!/bin/bash
# rocprofv3 kernel stats of the 3-D FlowNetS train step -> gpurun_out/prof3d/kernel_stats.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof3d
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3d/kt -- python3 $R/scratch/prof3d.py > $R/gpurun_out/prof3d/run.log 2>&1 || true
find $R/gpurun_out/prof3d/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof3d/kernel_stats.csv || true
cat $R/gpurun_out/prof3d/run.log | tail -5
head -24 $R/gpurun_out/prof3d/kernel_stats.csv | cut -c1-220
