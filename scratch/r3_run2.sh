#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15
python3 scratch/train_time.py 512 512 2>&1 | tail -12
python3 bench.py --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/r3_bench2.json; cut -c1-600 gpurun_out/r3_bench2.json
