#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -8
for extra in "" "--host-threads 4" "--host-threads 16"; do
  echo "== train_time 512 512 $extra"; python3 scratch/train_time.py 512 512 $extra 2>&1 | grep -E "INFO: lo|wall"
done
for extra in "--pingpong-min 100000" "--pingpong-min 512"; do
  echo "== train_time 1024 512 $extra"; python3 scratch/train_time.py 1024 512 $extra 2>&1 | grep -E "INFO: lo|wall"
done
python3 scratch/qb.py 2>&1 | tail -1
python3 scratch/qt.py 1 2048 24 2>&1 | tail -1
python3 scratch/qt.py 1 4096 64 2>&1 | tail -1
python3 scratch/qt.py 16 2048 24 2>&1 | tail -1
