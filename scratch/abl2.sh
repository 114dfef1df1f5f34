#!/bin/bash
R=$GRAFT_REPO_ROOT
for shape in "fwd 256 256 3 1 32 24" "fwd 128 256 5 2 64 24" "dgrad 256 256 3 1 32 24" "fwd 512 512 3 1 16 24" "fwd 64 128 5 2 128 24" "dgrad 128 256 5 2 64 24"; do
  for v in 0 64; do
    export MIREG_TILE_N=$v
    echo -n "[tile_n=$v] "; python3 $R/scratch/mb_conv.py $shape bf16 20 2>/dev/null | tail -1
  done
done
