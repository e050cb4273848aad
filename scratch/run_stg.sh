#!/bin/bash
for cfg in "0 0" "1 16" "1 32" "1 64" "1 128" "3 16" "3 32" "2 64" "7 8" "7 16"; do
  set -- $cfg
  TAG="mask=$1 unit=$2" MEDGP_CI_STG_MASK=$1 MEDGP_CI_STG_UNIT=$2 timeout 120 python3 scratch/quick_one.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
done
