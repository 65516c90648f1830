#!/bin/bash
# Same-box A/B of two library builds on the captured UNet step: ab_tmp/libsdamd_old.so vs the in-tree build, interleaved.
set -e
B=${1:-8}
for i in 1 2; do
  SDAMD_LIB=$PWD/ab_tmp/libsdamd_old.so timeout -k 10 200 python experiments/step_time.py $B $2
  timeout -k 10 200 python experiments/step_time.py $B $2
done
