#!/bin/bash
# Which side-stream kernel interferes with the upsampler backward-data kernel?  Each variant: two seeded trainers x 3 eager steps, 3 repeats.
export MIREG_TINY_MASK=7
for v in "all-on-side:" "up-on-main:up" "deconv-on-main:deconv" "pf-on-main:predict_flow" "deconv+pf-on-main:deconv,predict_flow" "up+pf-on-main:up,predict_flow" "up+deconv-on-main:up,deconv"; do
  tag=${v%%:*}; lst=${v#*:}
  res=""
  for i in 1 2 3; do
    out=$(MIREG_WGRAD_ON_MAIN=$lst timeout -k 10 100 python scratch/hazard_probe.py eager 2>&1 | grep "first step whose")
    res="$res | ${out#first step whose gradients differ: }"
  done
  echo "$tag (MIREG_WGRAD_ON_MAIN=$lst): first diverging step per run $res"
done
