#!/bin/bash
TAG=auto timeout 300 python3 scratch/quick_shapes.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
TAG=single44 MEDGP_MULTI_CU=-1 MEDGP_CHOLINV_NW=44 timeout 300 python3 scratch/quick_shapes.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
TAG=single84 MEDGP_MULTI_CU=-1 MEDGP_CHOLINV_NW=84 timeout 300 python3 scratch/quick_shapes.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
TAG=multi MEDGP_MULTI_CU=1 timeout 300 python3 scratch/quick_shapes.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
