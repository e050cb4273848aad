#!/bin/bash
for p in 0 16 32 48 80; do TAG="pad=$p" MEDGP_LD_PAD=$p timeout 120 python3 scratch/quick_one.py 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
