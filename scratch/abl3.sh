#!/bin/bash
R=$GRAFT_REPO_ROOT
for shape in "194 18 1 1 64 24" "386 18 1 1 32 24" "770 18 1 1 16 24" "1026 18 1 1 8 24" "1024 18 1 1 4 24"; do
  for mode in fwd dgrad wgrad; do
    python3 $R/scratch/mb_conv.py $mode $shape bf16 20 2>/dev/null | tail -1
  done
done
