#!/bin/bash
# A/B: thin_conv.hip compiled as shipped (53 packed-fp32 VALU instructions in the upsampler backward-data kernel) vs -fno-slp-vectorize (none)
export MIREG_TINY_MASK=7
for v in "packed-fp32-valu:" "no-packed-fp32:scratch/libmireg_nopk.so"; do
  tag=${v%%:*}; lib=${v#*:}
  res=""
  for i in 1 2 3 4 5 6; do
    out=$(ALT_LIB=$lib timeout -k 10 100 python scratch/hazard_probe.py eager 2>&1 | grep "first step whose")
    res="$res | ${out#first step whose gradients differ: }"
  done
  echo "$tag (ALT_LIB=$lib): first diverging step of two identically seeded trainers, six runs $res"
done
