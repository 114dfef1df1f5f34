#!/bin/bash
# A/B of the upsampler backward-data kernel: weights through 67 scalar loads inside the tap loop (as round 2 shipped it) vs staged once in LDS.
export MIREG_TINY_MASK=7
for v in "scalar-loads-in-loop:0" "weights-staged-in-LDS:16"; do
  tag=${v%%:*}; d=${v#*:}
  res=""
  for i in 1 2 3 4 5 6; do
    out=$(MIREG_HAZ_DBG=$d timeout -k 10 100 python scratch/hazard_probe.py eager 2>&1 | grep "first step whose")
    res="$res | ${out#first step whose gradients differ: }"
  done
  echo "$tag (MIREG_HAZ_DBG=$d): first diverging step of two identically seeded trainers, six runs $res"
done
