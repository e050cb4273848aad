#!/bin/bash
TAG=nw44 MEDGP_CHOLINV_NW=44 timeout 120 python3 scratch/quick_one.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
TAG=nw84 MEDGP_CHOLINV_NW=84 timeout 120 python3 scratch/quick_one.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
