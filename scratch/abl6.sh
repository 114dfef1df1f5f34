#!/bin/bash
R=$GRAFT_REPO_ROOT
for shape in "wgrad 256 256 3 1 32 24" "wgrad 128 256 5 2 64 24" "wgrad 512 512 3 1 16 24" "wgrad 64 128 5 2 128 24"; do
  unset MIREG_WG_STAGES3; echo -n "[4st] "; python3 $R/scratch/mb_conv.py $shape bf16 20 2>/dev/null | tail -1
  export MIREG_WG_STAGES3=1; echo -n "[3st] "; python3 $R/scratch/mb_conv.py $shape bf16 20 2>/dev/null | tail -1
done
