#!/bin/bash
# profile run on the GPU box: kernel trace + PMC passes (each in its own run, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/bench_${TAG}_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/bench_${TAG}_trace.log 2>&1
grep '^{' gpurun_out/bench_${TAG}_trace.log | cut -c1-400
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/bench_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/bench_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_${TAG}_mfma -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/bench_${TAG}_mfma.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/prof_${TAG}_sq -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/bench_${TAG}_sq.log 2>&1
find gpurun_out -name "*counter_collection.csv" | head; find gpurun_out/prof_${TAG}_trace -name "*kernel_stats.csv"
# round 6: the HBM side of two more shapes (verdict r5 item 6) -- the screening batch (1000 entries, nlml only) and BASELINE config 3 (1 x N = 2048):
# kernel trace + FETCH_SIZE / WRITE_SIZE passes each, same bench command with the shape's flags
for shp in "screen --patients 1000 --flag-grad 0" "cfg3 --patients 1 --obs 2048"; do
  set -- $shp; name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${name}_trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra "$@" > gpurun_out/bench_${TAG}_${name}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_${name}_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra "$@" > gpurun_out/bench_${TAG}_${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_${name}_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra "$@" > gpurun_out/bench_${TAG}_${name}_write.log 2>&1
done
