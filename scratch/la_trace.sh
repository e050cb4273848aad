#!/bin/bash
# per-launch durations of the look-ahead factorisation: rocprofv3 kernel trace of one evaluation
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=${1:-1}; N=${2:-2048}; D=${3:-24}
rm -rf gpurun_out/la_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/la_trace -- python3 scratch/qt.py $P $N $D > gpurun_out/la_trace.log 2>&1
tail -1 gpurun_out/la_trace.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/la_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last evaluation: find the last k_prep
idx = max(i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_prep'))
prev_end = None
out = []
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    out.append((r['Kernel_Name'][:22], (e - s) / 1e3, gap, r.get('Grid_Size_Y', r.get('Grid_Size', ''))))
    prev_end = e
tot = (int(rows[-1]['End_Timestamp']) - int(rows[idx]['Start_Timestamp'])) / 1e3
print('wall of last evaluation: %.1f us, launches %d' % (tot, len(out)))
for i, (n, d, g, gs) in enumerate(out):
    if i < 12 or i % 8 == 0 or i > len(out) - 6: print('%3d %-22s dur %7.1f us  gap %5.1f us  grid %s' % (i, n, d, g, gs))
PY
