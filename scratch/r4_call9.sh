#!/bin/bash
# round 4, GPU call 9: GPU suite incl. the SE / SM host test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c9
(time python3 -m pytest tests -m gpu -q) > gpurun_out/r4c9/pytest.log 2>&1
tail -30 gpurun_out/r4c9/pytest.log | cut -c1-300
