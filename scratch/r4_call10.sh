#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c10
(time python3 -m pytest tests/test_cohort_launchers_gpu.py -m gpu -q -x) > gpurun_out/r4c10/pytest.log 2>&1
tail -30 gpurun_out/r4c10/pytest.log | cut -c1-400
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
