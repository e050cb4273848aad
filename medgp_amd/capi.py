"""ctypes binding of libmedgp_hip.so (C ABI: include/medgp_hip.h).

Mirrors the reference's call shape: a Context plays the role of the (kernel, likelihood,
inference, prior) object set of main_one_train.cpp:103-152, `set_patient` of
c_objective_one's constructor, `nlml_grad` of c_objective_one::compute_objective.
Fails loudly when the HIP library is missing -- there is no CPU path.
"""
import ctypes as C
import weakref
import os

import numpy as np

KERNEL_SE, KERNEL_LMC_SM, KERNEL_SM = 0, 7, 8
PRIOR_NONE, PRIOR_CLAMP, PRIOR_NORMAL, PRIOR_LAPLACE = -1, 0, 1, 2
FLAG_GRAD, FLAG_KEEP_FACTOR = 1, 2   # flag_grad bits (include/medgp_hip.h)

_HERE = os.path.dirname(os.path.abspath(__file__))

# every symbol include/medgp_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "medgp_abi_version", "medgp_device_count", "medgp_create", "medgp_destroy", "medgp_last_error",
    "medgp_num_hyp", "medgp_set_pi", "medgp_set_stream", "medgp_reserve", "medgp_reserve_plan", "medgp_alloc_stats", "medgp_set_patient",
    "medgp_set_patients", "medgp_set_prior", "medgp_set_priors", "medgp_host_alloc", "medgp_host_free", "medgp_nlml_grad_async",
    "medgp_wait", "medgp_nlml_grad", "medgp_screen", "medgp_nlml_grad_device", "medgp_get_factor",
    "medgp_factor", "medgp_factor_batch", "medgp_pin_route", "medgp_last_plan", "medgp_fit_predict", "medgp_fit_predict_batch", "medgp_synchronize", "medgp_profile_enable", "medgp_profile_num_kernels",
    "medgp_profile_kernel_name", "medgp_profile_read", "medgp_profile_reset", "medgp_kde_mode", "medgp_kde_mode_at",
]


class MedgpError(RuntimeError):
    pass


def lib_path():
    """The in-tree library; MEDGP_LIB selects another build of the SAME ABI (A/B measurements of kernel variants)."""
    return os.environ.get("MEDGP_LIB") or os.path.join(_HERE, "libmedgp_hip.so")


_lib = None


def load():
    """dlopen the HIP library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise MedgpError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(make -C medgp_amd/csrc). medgp_amd has no CPU fallback.")
    # Share ONE HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's);
    # whichever is loaded first is used by both, and a torch initialised after a foreign runtime can come
    # up with "No HIP GPUs are available".  So when torch is importable, let it load first.  (The C ABI
    # itself has no torch dependency; C/C++ hosts are unaffected.)
    try:
        import torch  # noqa: F401
    except Exception:   # pragma: no cover
        pass
    lib = C.CDLL(p)
    vp, i32p, dp, fp, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    lib.medgp_abi_version.restype = C.c_int
    lib.medgp_device_count.restype = C.c_int
    lib.medgp_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.medgp_destroy.argtypes = [vp]
    lib.medgp_destroy.restype = None
    lib.medgp_last_error.argtypes = [vp]
    lib.medgp_last_error.restype = C.c_char_p
    lib.medgp_num_hyp.argtypes = [vp]
    lib.medgp_set_pi.argtypes = [vp, C.c_double]
    lib.medgp_set_stream.argtypes = [vp, vp]
    lib.medgp_reserve.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.medgp_reserve_plan.argtypes = [vp, C.c_int, i32p, C.c_int]
    lib.medgp_alloc_stats.argtypes = [vp, dp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.medgp_set_patient.argtypes = [vp, C.c_int, C.c_int, i32p, fp, fp]
    lib.medgp_set_patients.argtypes = [vp, C.c_int, i32p, C.POINTER(C.c_int64), i32p, fp, fp]
    lib.medgp_set_prior.argtypes = [vp, C.c_int, u8p, i32p, u8p, fp, fp]
    lib.medgp_set_priors.argtypes = [vp, C.c_int, i32p, u8p, i32p, u8p, fp, fp]
    lib.medgp_host_alloc.argtypes = [C.c_size_t]
    lib.medgp_host_alloc.restype = vp
    lib.medgp_host_free.argtypes = [vp]
    lib.medgp_host_free.restype = None
    lib.medgp_nlml_grad_async.argtypes = [vp, C.c_int, C.c_int, i32p, vp, C.c_int, vp, vp, vp]
    lib.medgp_wait.argtypes = [vp, C.c_int]
    lib.medgp_nlml_grad.argtypes = [vp, C.c_int, i32p, dp, C.c_int, dp, dp, i32p]
    lib.medgp_screen.argtypes = [vp, C.c_int, i32p, C.c_int, dp, dp, i32p]
    lib.medgp_nlml_grad_device.argtypes = [vp, C.c_int, i32p, vp, C.c_int, vp, vp, vp]
    lib.medgp_get_factor.argtypes = [vp, C.c_int, fp, fp, fp]
    lib.medgp_factor.argtypes = [vp, C.c_int, dp, dp, dp, i32p]
    lib.medgp_factor_batch.argtypes = [vp, C.c_int, i32p, dp, C.POINTER(dp), C.POINTER(dp), i32p]
    lib.medgp_pin_route.argtypes = [vp, C.c_int]
    lib.medgp_last_plan.argtypes = [vp, C.c_int, i32p, i32p, i32p]
    lib.medgp_fit_predict.argtypes = [vp, C.c_int, dp, C.c_int, i32p, fp, fp, fp, i32p]
    lib.medgp_fit_predict_batch.argtypes = [vp, C.c_int, i32p, dp, i32p, fp, fp, fp, i32p]
    lib.medgp_synchronize.argtypes = [vp]
    lib.medgp_profile_enable.argtypes = [vp, C.c_int]
    lib.medgp_profile_num_kernels.restype = C.c_int
    lib.medgp_profile_kernel_name.argtypes = [C.c_int]
    lib.medgp_profile_kernel_name.restype = C.c_char_p
    lib.medgp_profile_read.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_int64)]
    lib.medgp_profile_reset.argtypes = [vp]
    lib.medgp_kde_mode.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), i32p, dp, C.c_int, dp, dp, i32p, dp]
    lib.medgp_kde_mode_at.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), i32p, dp, C.POINTER(C.c_int64), i32p, dp, C.c_int, dp, dp, i32p, dp]
    _lib = lib
    return lib


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def _pack(series):
    ns = len(series)
    cnt = np.array([0 if x is None else np.size(x) for x in series], dtype=np.int32)
    off = np.zeros(ns, dtype=np.int64)
    if ns:
        off[1:] = np.cumsum(cnt[:-1], dtype=np.int64)
    parts = [np.asarray(x, dtype=np.float64).ravel() for x in series if x is not None]
    data = np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros(0))
    return off, cnt, data


def kde_mode(series, weighted=True, device=0, full=False, test=None):
    """medgp_kde_mode[_at] over a list of 1-D sample arrays: the KDE "mode" of each (compute_kde + compute_mode of the
    reference, ref: medgpc/clustering/mode_estimate.py:438-450).  test: optional list (entries may be None) of evaluation
    grids, one per series -- the density is then evaluated there and the mode taken over the grid.  Returns modes
    [len(series)]; with full=True also (bandwidths, status, kernel milliseconds).  status -1 marks a series the reference's
    KDE fit raises on."""
    lib = load()
    ns = len(series)
    off, cnt, data = _pack(series)
    mode, bw, st, ms = np.full(ns, np.nan), np.full(ns, np.nan), np.zeros(ns, dtype=np.int32), C.c_double(0.0)
    if test is None:
        rc = lib.medgp_kde_mode(int(device), ns, _ptr(off, C.c_int64), _ptr(cnt, C.c_int32), _ptr(data, C.c_double),
                                1 if weighted else 0, _ptr(mode, C.c_double), _ptr(bw, C.c_double), _ptr(st, C.c_int32), C.byref(ms))
    else:
        assert len(test) == ns
        toff, tcnt, tdata = _pack(test)
        rc = lib.medgp_kde_mode_at(int(device), ns, _ptr(off, C.c_int64), _ptr(cnt, C.c_int32), _ptr(data, C.c_double),
                                   _ptr(toff, C.c_int64), _ptr(tcnt, C.c_int32), _ptr(tdata, C.c_double), 1 if weighted else 0,
                                   _ptr(mode, C.c_double), _ptr(bw, C.c_double), _ptr(st, C.c_int32), C.byref(ms))
    if rc != 0:
        raise MedgpError(f"medgp_kde_mode failed ({rc}): {lib.medgp_last_error(None).decode()}")
    return (mode, bw, st, ms.value) if full else mode


class Context:
    """One (device, covariance family) evaluation context."""

    def __init__(self, kernel_index=KERNEL_LMC_SM, Q=5, D=2, R=2, device=0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.medgp_create(C.byref(h), int(device), int(kernel_index), int(Q), int(D), int(R))
        if rc != 0:
            raise MedgpError(f"medgp_create failed ({rc}): {self._lib.medgp_last_error(None).decode()}")
        self._h = h
        self.kernel_index, self.Q, self.D, self.R, self.device = kernel_index, Q, D, R, device
        self.H = self._lib.medgp_num_hyp(h)

    def _chk(self, rc):
        if rc != 0:
            raise MedgpError(f"libmedgp_hip error {rc}: {self._lib.medgp_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.medgp_destroy(self._h)
            self._h = None
            self._lane_refs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_pi(self, pi):
        self._chk(self._lib.medgp_set_pi(self._h, float(pi)))

    def set_stream(self, stream_ptr):
        self._chk(self._lib.medgp_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def reserve(self, max_slots, max_n, max_batch):
        self._chk(self._lib.medgp_reserve(self._h, int(max_slots), int(max_n), int(max_batch)))

    def reserve_plan(self, sizes, ninit=0):
        """medgp_reserve_plan: announce the sizes of the patients that will be resident together (and the width of the screening):
        the per-entry arenas are allocated once to what the largest call over them needs."""
        sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        self._chk(self._lib.medgp_reserve_plan(self._h, int(sizes.shape[0]), _ptr(sizes, C.c_int32), int(ninit)))

    def alloc_stats(self):
        """(seconds spent in device-memory management calls, number of such calls, bytes held by the per-entry arenas)"""
        s, n, b = C.c_double(0.0), C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.medgp_alloc_stats(self._h, C.byref(s), C.byref(n), C.byref(b)))
        return s.value, n.value, b.value

    def set_patient(self, slot, meta, t, y):
        t = np.ascontiguousarray(t, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        meta = None if meta is None else np.ascontiguousarray(meta, dtype=np.int32)
        self._chk(self._lib.medgp_set_patient(self._h, int(slot), int(t.shape[0]), _ptr(meta, C.c_int32),
                                              _ptr(t, C.c_float), _ptr(y, C.c_float)))

    def set_patients(self, slots, patients):
        """Packed upload: patients = list of (meta, t, y); one H2D transfer, no device wait."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        ns = [np.asarray(p[1]).shape[0] for p in patients]
        offsets = np.zeros(len(ns) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum(ns)
        t = np.ascontiguousarray(np.concatenate([np.asarray(p[1], dtype=np.float32) for p in patients]), dtype=np.float32)
        y = np.ascontiguousarray(np.concatenate([np.asarray(p[2], dtype=np.float32) for p in patients]), dtype=np.float32)
        meta = None
        if patients[0][0] is not None:
            meta = np.ascontiguousarray(np.concatenate([np.asarray(p[0], dtype=np.int32) for p in patients]), dtype=np.int32)
        self._chk(self._lib.medgp_set_patients(self._h, len(ns), _ptr(slots, C.c_int32), offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                               _ptr(meta, C.c_int32), _ptr(t, C.c_float), _ptr(y, C.c_float)))

    def set_prior(self, slot, flag=None, type=None, is_exp=None, p0=None, p1=None):
        if flag is None:
            self._chk(self._lib.medgp_set_prior(self._h, int(slot), None, None, None, None, None))
            return
        flag = np.ascontiguousarray(flag, dtype=np.uint8)
        type = np.ascontiguousarray(type, dtype=np.int32)
        is_exp = np.ascontiguousarray(is_exp, dtype=np.uint8)
        p0 = np.ascontiguousarray(p0, dtype=np.float32)
        p1 = np.ascontiguousarray(p1, dtype=np.float32)
        assert flag.shape[0] == self.H
        self._chk(self._lib.medgp_set_prior(self._h, int(slot), _ptr(flag, C.c_uint8), _ptr(type, C.c_int32),
                                            _ptr(is_exp, C.c_uint8), _ptr(p0, C.c_float), _ptr(p1, C.c_float)))

    def set_priors(self, slots, flag=None, type=None, is_exp=None, p0=None, p1=None):
        """Batched medgp_set_prior: arrays are [len(slots), H]; one transfer, no device wait."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        ns = slots.shape[0]
        if flag is None:
            self._chk(self._lib.medgp_set_priors(self._h, ns, _ptr(slots, C.c_int32), None, None, None, None, None))
            return
        arrs = [np.ascontiguousarray(a, dtype=dt).reshape(ns, self.H) for a, dt in
                ((flag, np.uint8), (type, np.int32), (is_exp, np.uint8), (p0, np.float32), (p1, np.float32))]
        self._chk(self._lib.medgp_set_priors(self._h, ns, _ptr(slots, C.c_int32), _ptr(arrs[0], C.c_uint8), _ptr(arrs[1], C.c_int32),
                                             _ptr(arrs[2], C.c_uint8), _ptr(arrs[3], C.c_float), _ptr(arrs[4], C.c_float)))

    def pinned(self, shape, dtype):
        """numpy array in pinned host memory (medgp_host_alloc).  The memory belongs to the ARRAY, not to the context: it is
        freed (medgp_host_free) when the last view of it is garbage collected, so an array a caller still holds after close()
        stays valid.  Arrays handed to nlml_grad_async are additionally kept alive by the context until wait(lane)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self._lib.medgp_host_alloc(max(n, 1))
        if not p:
            raise MedgpError("medgp_host_alloc failed")
        buf = (C.c_char * max(n, 1)).from_address(p)
        weakref.finalize(buf, self._lib.medgp_host_free, p)   # every numpy view keeps `buf` alive through its .base chain
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def nlml_grad_async(self, lane, slots, theta, flag_grad, nlml, grad, status):
        """medgp_nlml_grad_async: theta / nlml / grad / status are (pinned) numpy arrays that stay alive until wait(lane)."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        if not hasattr(self, "_lane_refs"):
            self._lane_refs = {}
        self._lane_refs[int(lane)] = (theta, nlml, grad, status)   # the device writes into these until wait(lane)
        self._chk(self._lib.medgp_nlml_grad_async(self._h, int(lane), slots.shape[0], _ptr(slots, C.c_int32),
                                                  C.c_void_p(theta.ctypes.data), int(bool(flag_grad)), C.c_void_p(nlml.ctypes.data),
                                                  C.c_void_p(grad.ctypes.data if grad is not None else 0),
                                                  C.c_void_p(status.ctypes.data if status is not None else 0)))

    def wait(self, lane):
        self._chk(self._lib.medgp_wait(self._h, int(lane)))
        getattr(self, "_lane_refs", {}).pop(int(lane), None)

    def pin_route(self, pinned=True):
        """medgp_pin_route: one factorisation kernel for every call, so a patient's bits do not depend on its batch-mates."""
        self._chk(self._lib.medgp_pin_route(self._h, 1 if pinned else 0))

    def screen(self, slots, theta):
        """medgp_screen: the hyper vectors theta [ninit, H] evaluated (nlml only) on every patient of slots. Returns (nlml, status),
        both [len(slots), ninit]."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(-1, self.H)
        ns, ni = slots.shape[0], theta.shape[0]
        nlml = np.empty((ns, ni), dtype=np.float64)
        st = np.empty((ns, ni), dtype=np.int32)
        self._chk(self._lib.medgp_screen(self._h, ns, _ptr(slots, C.c_int32), ni, _ptr(theta, C.c_double), _ptr(nlml, C.c_double), _ptr(st, C.c_int32)))
        return nlml, st

    def last_plan(self):
        """medgp_last_plan: [(entries, 64-blocks of the largest, route)] of the last nlml_grad call's size classes, largest first;
        route 0 / 1 = one workgroup per entry (4- / 8-wave shape), 2 = multi-CU look-ahead schedule."""
        cnt, blk, rt = (np.zeros(32, np.int32) for _ in range(3))
        nc = self._lib.medgp_last_plan(self._h, 32, _ptr(cnt, C.c_int32), _ptr(blk, C.c_int32), _ptr(rt, C.c_int32))
        if nc < 0:
            self._chk(nc)
        return [(int(cnt[i]), int(blk[i]), int(rt[i])) for i in range(min(nc, 32))]

    def nlml_grad(self, slots, theta, flag_grad=True, keep_factor=False):
        """Host-pointer operator. theta: [nbatch, H]. Returns (nlml[nbatch], grad[nbatch,H] or None, status[nbatch]).
        keep_factor: MEDGP_FLAG_KEEP_FACTOR (alpha / L^-1 available through get_factor even without gradients)."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(slots.shape[0], self.H)
        nb = slots.shape[0]
        nlml = np.empty(nb)
        grad = np.empty((nb, self.H)) if flag_grad else None
        status = np.empty(nb, dtype=np.int32)
        self._chk(self._lib.medgp_nlml_grad(self._h, nb, _ptr(slots, C.c_int32), _ptr(theta, C.c_double),
                                            int(bool(flag_grad)) | (FLAG_KEEP_FACTOR if keep_factor else 0),
                                            _ptr(nlml, C.c_double), _ptr(grad, C.c_double), _ptr(status, C.c_int32)))
        return nlml, grad, status

    def nlml_grad_device(self, slots, theta_ptr, flag_grad, nlml_ptr, grad_ptr, status_ptr):
        """Device-pointer operator (asynchronous on the context's stream). Pointers are integers (tensor.data_ptr())."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        self._chk(self._lib.medgp_nlml_grad_device(self._h, slots.shape[0], _ptr(slots, C.c_int32), C.c_void_p(theta_ptr),
                                                   int(bool(flag_grad)), C.c_void_p(nlml_ptr), C.c_void_p(grad_ptr or 0),
                                                   C.c_void_p(status_ptr or 0)))

    def get_factor(self, b, n, want_linv=True):
        alpha = np.empty(n, dtype=np.float32)
        linv = np.empty((n, n), dtype=np.float32) if want_linv else None
        beta = C.c_float()
        self._chk(self._lib.medgp_get_factor(self._h, int(b), _ptr(alpha, C.c_float), _ptr(linv, C.c_float), C.byref(beta)))
        return alpha, linv, beta.value

    def factor(self, slot, theta, n):
        """Cholesky factor (caller order, lower, fp64) and z = L^-1 y of one patient. Returns (L[n,n], z[n], status)."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        Lm = np.zeros((n, n))
        z = np.zeros(n)
        st = C.c_int32()
        self._chk(self._lib.medgp_factor(self._h, int(slot), _ptr(theta, C.c_double), _ptr(Lm, C.c_double), _ptr(z, C.c_double), C.byref(st)))
        return Lm, z, st.value

    def factor_batch(self, slots, theta, ns):
        """medgp_factor_batch: list of (L[n,n], z[n]) per entry and the status array; ns = observations of each entry."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        nb = slots.shape[0]
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(nb, self.H)
        Ls = [np.zeros((int(n), int(n))) for n in ns]
        zs = [np.zeros(int(n)) for n in ns]
        Lp = (C.POINTER(C.c_double) * nb)(*[_ptr(a, C.c_double) for a in Ls])
        zp = (C.POINTER(C.c_double) * nb)(*[_ptr(a, C.c_double) for a in zs])
        st = np.zeros(nb, dtype=np.int32)
        self._chk(self._lib.medgp_factor_batch(self._h, nb, _ptr(slots, C.c_int32), _ptr(theta, C.c_double), Lp, zp, _ptr(st, C.c_int32)))
        return list(zip(Ls, zs)), st

    def fit_predict(self, slot, theta, meta2, t2):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        t2 = np.ascontiguousarray(t2, dtype=np.float32)
        meta2 = None if meta2 is None else np.ascontiguousarray(meta2, dtype=np.int32)
        ns = t2.shape[0]
        mean = np.empty(ns, dtype=np.float32)
        var = np.empty(ns, dtype=np.float32)
        st = C.c_int32()
        self._chk(self._lib.medgp_fit_predict(self._h, int(slot), _ptr(theta, C.c_double), ns, _ptr(meta2, C.c_int32),
                                              _ptr(t2, C.c_float), _ptr(mean, C.c_float), _ptr(var, C.c_float), C.byref(st)))
        return mean, var, st.value

    def fit_predict_batch(self, slots, theta, meta2, t2):
        """One test point per problem: problem b = (slots[b], theta[b], meta2[b], t2[b])."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        nb = slots.shape[0]
        theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(nb, self.H)
        t2 = np.ascontiguousarray(t2, dtype=np.float32)
        meta2 = None if meta2 is None else np.ascontiguousarray(meta2, dtype=np.int32)
        mean = np.empty(nb, dtype=np.float32)
        var = np.empty(nb, dtype=np.float32)
        st = np.empty(nb, dtype=np.int32)
        self._chk(self._lib.medgp_fit_predict_batch(self._h, nb, _ptr(slots, C.c_int32), _ptr(theta, C.c_double),
                                                    _ptr(meta2, C.c_int32), _ptr(t2, C.c_float), _ptr(mean, C.c_float),
                                                    _ptr(var, C.c_float), _ptr(st, C.c_int32)))
        return mean, var, st

    def synchronize(self):
        self._chk(self._lib.medgp_synchronize(self._h))

    # measurement hooks
    def profile_enable(self, on=True, only=None):
        """on: bracket every launch with HIP events; only='k_cholinv': just the launches of that kernel."""
        mode = int(bool(on))
        if on and only is not None:
            names = [self._lib.medgp_profile_kernel_name(k).decode() for k in range(self._lib.medgp_profile_num_kernels())]
            mode = 2 + names.index(only)
        self._chk(self._lib.medgp_profile_enable(self._h, mode))

    def profile_reset(self):
        self._chk(self._lib.medgp_profile_reset(self._h))

    def profile_read(self):
        out = {}
        for k in range(self._lib.medgp_profile_num_kernels()):
            ms, cnt = C.c_double(), C.c_int64()
            self._chk(self._lib.medgp_profile_read(self._h, k, C.byref(ms), C.byref(cnt)))
            out[self._lib.medgp_profile_kernel_name(k).decode()] = (ms.value, cnt.value)
        return out
