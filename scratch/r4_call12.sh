#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/dbg.py <<'PY'
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
os.environ['MEDGP_DBG_NOWGRAD']='1'
import medgp_amd
from medgp_amd import capi, synth
capi.lib_path = lambda: os.environ['STAMP_LIB']
D,N,Q,R=24,512,5,8
P=int(os.environ.get("SP","256"))
pts, th = synth.cohort(11, 16, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
ctx.set_patients(np.arange(P), [pts[s % 16] for s in range(P)])
th = np.stack([th[s % 16] for s in range(P)])
lib=capi.load()
print(capi.lib_path(), hasattr(lib, 'medgp_debug_read_diag'))
buf=np.zeros(8,np.uint64)
lib.medgp_debug_read_diag.argtypes=[C.c_void_p]
for it in range(3):
    nl,g,st=ctx.nlml_grad(np.arange(P), th, True)
    rc=lib.medgp_debug_read_diag(buf.ctypes.data_as(C.c_void_p))
    print(it, rc, st[:4], buf)
a=buf.astype(np.float64); n=max(a[7],1)
names=['diag16 x4','panel tiles','trailing tiles','inverse tiles','zero+log']
print(f"P={P}: {int(n)} block factorisations; cycles per factorisation:", {names[i]: int(a[i]/n) for i in range(5)}, 'total', int(a[:5].sum()/n))
PY
for SP in 256 512; do for v in old new; do echo "== $v SP=$SP"; SP=$SP STAMP_LIB=/root/repo/scratch/libmedgp_stamps_$v.so MEDGP_MULTI_CU=-1 MEDGP_CHOLINV_NW=44 python3 /tmp/dbg.py 2>&1 | grep -v amdgpu | tail -5; done; done
