#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in lib_lastamps.so lib_lastamps_pair.so; do echo "=== $l"; LASTAMP_LIB=$l timeout 300 python3 scratch/la_stamps.py 2048 24 2>&1 | grep -v amdgpu | tail -30; done
