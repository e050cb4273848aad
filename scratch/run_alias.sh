#!/bin/bash
timeout 120 python3 scratch/quick_lib.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
LIB=/root/repo/scratch/libmedgp_alias512.so timeout 120 python3 scratch/quick_lib.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
LIB=/root/repo/scratch/libmedgp_alias4.so timeout 120 python3 scratch/quick_lib.py 2>&1 | grep -v -i "warn\|amdgpu.ids"
