#!/bin/bash
cd $GRAFT_REPO_ROOT
for pf in 1 2; do echo "=== pf $pf"; MEDGP_WGRAD_DEEP=$pf timeout 200 python3 scratch/wg_trace.py 2048 24 2>&1 | grep -v amdgpu | tail -40; done
