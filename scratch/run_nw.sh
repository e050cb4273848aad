#!/bin/bash
for nw in 44 82 44 82; do TAG=nw$nw MEDGP_CHOLINV_NW=$nw timeout 120 python3 scratch/quick_one.py 2>&1 | grep -v -i "warn\|amdgpu.ids"; done
