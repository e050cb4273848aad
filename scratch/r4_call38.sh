#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c38
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4c38/bench_driver_cmd.log 2>&1
tail -1 gpurun_out/r4c38/bench_driver_cmd.log | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac']); oc=d['other_configs']
for k in ('train_cohort_512xN512_D24','test_cohort_64xN120-200_D4','host_paths'):
    if k in oc: print(k, oc[k])"
timeout 600 python -m pytest tests/test_test_host_gpu.py tests/test_train_host_gpu.py -q -m gpu -x 2>&1 | tail -3
