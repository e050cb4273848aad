#!/bin/bash
# round 4, GPU call 20: final sources: GPU suite, stamps, bench line, profile set (kernel trace + PMC), bench line again with current traffic
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c20
O=gpurun_out/r4c20
(time python3 -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head
python3 scratch/la_stamps.py 2048 24 > $O/stamps_2048.log 2>&1
python3 scratch/la_stamps.py 4096 64 > $O/stamps_4096.log 2>&1
bash scratch/gpurun_prof.sh r04 2>&1 | tail -2
