#!/bin/bash
# medgp_screen on the large patients of the heavy-tailed cohort for different chunk-closing thresholds (MEDGP_SCREEN_WORK, block pairs)
cd $GRAFT_REPO_ROOT
for w in 4096 8192 16384 32768 65536; do
  echo "== MEDGP_SCREEN_WORK=$w"
  MEDGP_SCREEN_WORK=$w python3 scratch/screen_ragged.py 2>&1 | grep -v amdgpu | head -2
done
