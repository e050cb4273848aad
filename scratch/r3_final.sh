#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -m gpu -q 2>&1 | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep '^{' > gpurun_out/bench_r03_full.json; cut -c1-300 gpurun_out/bench_r03_full.json
bash scratch/gpurun_prof.sh r03 > gpurun_out/r3_prof.log 2>&1; tail -3 gpurun_out/r3_prof.log
python3 scratch/train_time.py 512 512 2>&1 | grep -E "INFO: lo|wall"
