#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -15
for extra in "" "--host-threads 8" "--host-threads 4" "--pingpong-min 256" "--pingpong-min 256 --host-threads 8" "--pingpong-min 128"; do
  echo "== train_time 512 512 $extra"; python3 scratch/train_time.py 512 512 $extra 2>&1 | grep -E "INFO: lo|wall"
done
