#!/bin/bash
# After the fix (library built without packed-fp32 VALU ops, all upsampler kernels on): two identically seeded trainers, six runs each, eager and hipGraph
for mode in eager graph; do
  res=""
  for i in 1 2 3 4 5 6; do
    out=$(timeout -k 10 100 python scratch/hazard_probe.py $mode 2>&1 | grep "first step whose")
    res="$res | ${out#first step whose gradients differ: }"
  done
  echo "$mode, MIREG_TINY_MASK default (7): first diverging step of two identically seeded trainers, six runs $res"
done
