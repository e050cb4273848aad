#!/bin/bash
# round 4, GPU call 8: bench line of the final sources (traffic provenance "current"), strong-scaling mode at N=1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c8
O=gpurun_out/r4c8
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
tail -1 $O/bench_driver_cmd.log | cut -c1-200
python3 bench.py --gpus 1 --steps 5 --warmup 2 --scaling strong --no-extra --no-cpu-baseline > $O/bench_strong.log 2>&1
tail -1 $O/bench_strong.log | cut -c1-400
