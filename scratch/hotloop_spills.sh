#!/bin/bash
# compile medgp_capi.hip to ISA and report scratch ops per barrier-delimited block of k_cholinv<4>
cd /root/repo/medgp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 "$@" -S --cuda-device-only medgp_capi.hip -o /tmp/hl.s 2>/dev/null
awk '/^_Z9k_cholinvILi4ELi4ELi0EEv8MedgpDevi:/,/s_endpgm/' /tmp/hl.s > /tmp/hl4.s
python3 - <<'PY'
lines=open('/tmp/hl4.s').read().split('\n')
idx=[0]+[i for i,l in enumerate(lines) if 's_barrier' in l]+[len(lines)]
print("lines",len(lines),"total scratch ops",sum('scratch_' in l for l in lines),"mfma",sum('v_mfma' in l for l in lines))
for a,b in zip(idx,idx[1:]):
    seg=lines[a:b]
    n=sum('v_mfma' in l for l in seg); sc=sum('scratch_' in l for l in seg)
    if n or sc: print(f"  block {a}-{b}: mfma {n}, scratch {sc}, vmcnt(0) {sum('vmcnt(0)' in l for l in seg)}, vmem loads {sum(('global_load' in l) or ('flat_load' in l) for l in seg)}, stores {sum(('global_store' in l) for l in seg)}")
PY
grep -A12 "\.name:  *_Z9k_cholinvILi4ELi4ELi0EEv8MedgpDevi" /tmp/hl.s | grep -E "vgpr_spill|private_segment" 
