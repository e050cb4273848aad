"""Regenerate profiles/<round>_* from the rocprofv3 outputs of scratch/gpurun_prof.sh <tag> (gpurun_out/prof_<tag>_*).
usage: python scratch/make_profiles.py <tag> [round, default r02].  Records the git commit and the digest of the device
sources the passes were taken from (bench.py refuses a stale traffic figure)."""
import csv, glob, collections, json, os, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
shutil.copy(glob.glob(f'gpurun_out/prof_{tag}_trace/runc/*_kernel_stats.csv')[0], f'profiles/{rnd}_kernel_stats.csv')
open(f'profiles/{rnd}_bench_line_under_rocprof.json', 'w').write([l for l in open(f'gpurun_out/bench_{tag}_trace.log') if l.startswith('{')][-1])
try:
    open(f'profiles/{rnd}_bench_line.json', 'w').write([l for l in open(f'gpurun_out/bench_{tag}_plain.log') if l.startswith('{')][-1])
except OSError:
    pass
# steady-state per-kernel durations from the kernel trace itself: the --stats summary averages the cold launches in (round 2: max
# 1.65 ms vs min 1.36 ms for k_cholinv), so the warm-up launches of every kernel are dropped here and the median is reported too
WARM = int(os.environ.get("PROFILE_WARMUP_LAUNCHES", "3"))   # = --warmup of the traced bench command (scratch/gpurun_prof.sh)
tr = glob.glob(f'gpurun_out/prof_{tag}_trace/runc/*_kernel_trace.csv')
if tr:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        per[r['Kernel_Name']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    with open(f'profiles/{rnd}_kernel_steady.csv', 'w') as f:
        f.write(f'"Name","LaunchesUsed","WarmupDropped","MedianNs","MeanNs","MinNs","MaxNs"\n')
        for k, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
            v.sort()
            d = [x[1] for x in v[min(WARM, max(len(v) - 1, 0)):]]
            d.sort()
            f.write(f'"{k}",{len(d)},{len(v) - len(d)},{d[len(d) // 2]},{sum(d) / len(d):.1f},{d[0]},{d[-1]}\n')
out = {}
for t in ('fetch', 'write', 'mfma', 'sq'):
    p = glob.glob(f'gpurun_out/prof_{tag}_{t}/runc/*_counter_collection.csv')[0]
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name']
        if k.startswith('__amd'):
            continue
        d[k.split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in d.items():
        out.setdefault(k, {}).update({c: {"mean_per_launch": sum(x) / len(x), "launches": len(x)} for c, x in v.items()})
for k, v in out.items():
    if k.startswith('_'): continue
    if 'FETCH_SIZE' in v:
        f = v['FETCH_SIZE']['mean_per_launch'] * 1024
        w = v['WRITE_SIZE']['mean_per_launch'] * 1024
        v['derived'] = {"fetch_bytes_raw": f, "fetch_bytes_x2_if_wide_loads": 2 * f, "write_bytes": w}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v:
        cyc = v['GRBM_GUI_ACTIVE']['mean_per_launch'] / 8.0
        v.setdefault('derived', {}).update({
            "kernel_cycles": cyc,
            "mfma_busy_frac_of_1024_simds": v['SQ_VALU_MFMA_BUSY_CYCLES']['mean_per_launch'] / (1024 * cyc),
            "mfma_f64_instructions": v['SQ_INSTS_VALU_MFMA_MOPS_F64']['mean_per_launch'] / 4.0,
            "valu_wave_instructions": v['SQ_INSTS_VALU']['mean_per_launch'],
            "valu_issue_frac_of_1024_simds": v['SQ_INSTS_VALU']['mean_per_launch'] * 4 / (1024 * cyc),
            "wait_any_frac_of_wave_cycles": v['SQ_WAIT_ANY']['mean_per_launch'] / v['SQ_WAVE_CYCLES']['mean_per_launch'] if 'SQ_WAVE_CYCLES' in v else None})
# round 6: other shapes (scratch/gpurun_prof.sh): steady-state launch time from the kernel trace + FETCH_SIZE / WRITE_SIZE per launch, per kernel
shapes = {}
for name, label in (("screen", "screening_1000xN512_nlml_only"), ("cfg3", "config3_1xN2048")):
    trs = glob.glob(f'gpurun_out/prof_{tag}_{name}_trace/runc/*_kernel_trace.csv')
    if not trs:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trs[0])):
        per[r['Kernel_Name'].split('(')[0].replace('void ', '')].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    cnt = {}
    for t in ('fetch', 'write'):
        ps = glob.glob(f'gpurun_out/prof_{tag}_{name}_{t}/runc/*_counter_collection.csv')
        if not ps:
            continue
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(ps[0])):
            if not r['Kernel_Name'].startswith('__amd'):
                d[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
        cnt[t] = {k: (sum(v) / len(v), len(v)) for k, v in d.items()}
    sh = {}
    for k, v in per.items():
        v.sort()
        d = sorted(x[1] for x in v[min(len(v) - 1, len(v) // 4):])       # the first quarter of the launches are warm-up steps
        e = {"median_ns": d[len(d) // 2], "mean_ns": sum(d) / len(d), "launches_used": len(d)}
        if k in cnt.get('fetch', {}) and k in cnt.get('write', {}):
            f, w = cnt['fetch'][k][0] * 1024, cnt['write'][k][0] * 1024
            e.update({"fetch_bytes_raw": f, "fetch_bytes_x2_if_wide_loads": 2 * f, "write_bytes": w})
        sh[k] = e
    shapes[label] = sh
    shutil.copy(glob.glob(f'gpurun_out/prof_{tag}_{name}_trace/runc/*_kernel_stats.csv')[0], f'profiles/{rnd}_kernel_stats_{name}.csv')
if shapes:
    out['_shapes'] = shapes
import bench
out['_meta'] = {'git_commit': subprocess.run(['git', 'rev-parse', 'HEAD'], capture_output=True, text=True).stdout.strip(),
                'csrc_sha256': bench.csrc_digest(), 'command': 'python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra (one rocprofv3 --pmc pass per counter group)'}
json.dump(out, open(f'profiles/{rnd}_pmc_summary.json', 'w'), indent=1)
for k, v in out.items():
    if k.startswith('_'): continue
    print(k, {a: (round(b, 3) if isinstance(b, float) and b < 10 else int(b)) for a, b in v.get('derived', {}).items()})
print(open(f'profiles/{rnd}_kernel_stats.csv').read()[:1100])
if tr:
    print(open(f'profiles/{rnd}_kernel_steady.csv').read()[:1100])
