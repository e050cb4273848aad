#!/bin/bash
for l in "$@"; do LIB=/root/repo/scratch/$l timeout 120 python3 scratch/quick_lib.py 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
