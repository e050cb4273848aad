import os, subprocess
print('nproc', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, 'n/a')
print(subprocess.run('lscpu | head -20; free -g | head -2', shell=True, capture_output=True, text=True).stdout)
