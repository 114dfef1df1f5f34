#!/bin/bash
R=$GRAFT_REPO_ROOT
for shape in "fwd 256 256 3 1 32 24" "dgrad 256 256 3 1 32 24" "fwd 512 512 3 1 16 24" "fwd 512 512 3 1 8 24" "fwd 1024 1024 3 1 4 24" "fwd 128 256 5 2 64 24" "dgrad 128 256 5 2 64 24"; do
  for v in 0 3; do
    export MIREG_STAGES=$v
    echo -n "[stages=$v] "; python3 $R/scratch/mb_conv.py $shape bf16 20 2>/dev/null | tail -1
  done
done
