#!/bin/bash
# The round's evidence in one box session: default bench line, kernel-trace stats of the same command, PMC passes.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ev
cd $R
timeout -k 10 700 python3 bench.py > gpurun_out/ev/bench_default.json 2> gpurun_out/ev/bench_default.err || exit 1
echo "bench done"
bash scratch/prof.sh ev_kt --no-3d > gpurun_out/ev/prof.txt 2>&1
echo "kernel trace done"
