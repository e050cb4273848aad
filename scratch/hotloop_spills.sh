#!/bin/bash
# compile medgp_capi.hip to ISA and report scratch ops inside the main MFMA loop of k_cholinv<4,4>
cd /root/repo/medgp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 "$@" -S --cuda-device-only medgp_capi.hip -o /tmp/hl.s 2>/dev/null
awk '/^_Z9k_cholinvILi4ELi4EEv8MedgpDevi:/,/s_endpgm/' /tmp/hl.s > /tmp/hl44.s
python3 - <<'PY'
import re
lines=open('/tmp/hl44.s').read().split('\n')
# find the first block with >= 60 mfma between two s_barrier
idx=[i for i,l in enumerate(lines) if 's_barrier' in l]
best=None
for a,b in zip(idx,idx[1:]):
    seg=lines[a:b]
    n=sum('v_mfma' in l for l in seg)
    if n>=60:
        sc=sum('scratch_' in l for l in seg); vm0=sum('vmcnt(0)' in l for l in seg)
        print(f"hot loop block lines {a}-{b}: mfma {n}, scratch ops {sc}, vmcnt(0) waits {vm0}, global loads {sum('global_load' in l for l in seg)}")
        break
PY
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size):" /tmp/hl.s | head -0
