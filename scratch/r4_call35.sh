#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c35
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4c35/bench_driver_cmd.log 2>&1
tail -1 gpurun_out/r4c35/bench_driver_cmd.log | cut -c1-200
