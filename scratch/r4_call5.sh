#!/bin/bash
# round 4, GPU call 5: full GPU suite on the cleaned sources, then the profile set of the driver's bench command (kernel trace + PMC passes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c5
O=gpurun_out/r4c5
(time python3 -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
tail -1 $O/bench_driver_cmd.log | cut -c1-300
bash scratch/gpurun_prof.sh r04 2>&1 | tail -5
