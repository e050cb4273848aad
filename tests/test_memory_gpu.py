"""Device memory of a context (round 6): per-entry buffers sized ONCE where the sizes are known (medgp_reserve_plan, from the patients'
sizes), grown without any wait where they are not (outgrown blocks are retired and freed at the next idle point), nlml-only calls that never touch Linv, and memory WAVES for calls whose
per-entry matrices exceed the budget.  None of it may change a bit of the results: every check here is against the same call on a
context that took the other path, bit for bit (one factorisation schedule pinned where the comparison crosses batch compositions),
and against the oracle at the parity bars of test_parity_gpu.py.
ref for the behaviour being replaced: the reference news / deletes its N x N buffers per evaluation (core/gp_regression.cpp:102-117,
inference/c_inference_exact.cpp:66-68,168)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import medgp_amd
from medgp_amd import capi, synth
from oracle import oracle as O

D, Q, R = 6, 3, 2
SIZES = [700, 40, 333, 129, 64, 512, 65, 200, 1100, 90, 18, 260, 131, 450, 77, 300]


def _cohort(sizes, seed=11):
    pts = [synth.patient(seed, p, D, int(n)) for p, n in enumerate(sizes)]
    th = np.stack([synth.theta(seed, p, 7, Q, D, R) for p in range(len(sizes))])
    return pts, th


def _ctx(sizes, pts, max_batch=None, pin=False):
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(len(sizes), int(max(sizes)), max_batch or len(sizes))
    ctx.set_patients(np.arange(len(sizes)), pts)
    ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    if pin:
        ctx.pin_route(True)
    return ctx


def test_reserve_plan_maps_once_and_changes_no_bit(monkeypatch):
    # capacities far above 8 GB of matrices (64 entries x N = 6000): the arenas start empty and would grow with the calls
    sizes = SIZES
    pts, th = _cohort(sizes)
    big = 6000
    slots = np.arange(len(sizes))
    inits = np.stack([synth.theta(5, s, 7, Q, D, R) for s in range(7)])

    def run(plan):
        ctx = medgp_amd.Context(7, Q, D, R)
        ctx.reserve(len(sizes), big, 64)
        ctx.set_patients(slots, pts)
        ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
        if plan:
            ctx.reserve_plan(sizes, ninit=inits.shape[0])
        s0 = ctx.alloc_stats()
        a = ctx.nlml_grad(slots, th, True)
        b = ctx.screen(slots, inits)
        c = ctx.nlml_grad(slots[::-1].copy(), th[::-1].copy(), False)
        s1 = ctx.alloc_stats()
        ctx.close()
        return a, b, c, s0, s1

    a0, b0, c0, s00, s01 = run(False)
    a1, b1, c1, s10, s11 = run(True)
    # same bits either way
    for x, y in zip(a0 + b0 + c0, a1 + b1 + c1):
        if x is not None:
            assert np.array_equal(x, y)
    assert (a0[2] == 0).all() and (b0[1] == 0).all()
    # with the plan announced the calls obtain no memory; without it they had to grow the buffers
    assert s11[2] == s10[2] and s11[1] - s10[1] <= 1, (s10, s11)      # (one call: medgp_screen's block of hyper vectors)
    assert s01[2] > s00[2]
    # and the plan is tight: never more than what the calls alone ended up with (they grow by 1.5 x, the plan allocates the exact
    # high-water mark of the same layouts)
    assert s11[2] <= s01[2], (s11, s01)
    assert s11[2] < 2 ** 30            # ~ 100 MB for this cohort, not max_batch x max_n^2 x 16 B = 37 GB


def test_nlml_only_calls_do_not_map_linv():
    sizes = SIZES
    pts, th = _cohort(sizes)
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(len(sizes), 6000, 64)            # lazy arenas
    ctx.set_patients(np.arange(len(sizes)), pts)
    b0 = ctx.alloc_stats()[2]
    nl0, _, st0 = ctx.nlml_grad(np.arange(len(sizes)), th, False)
    b1 = ctx.alloc_stats()[2]
    nl1, g1, st1 = ctx.nlml_grad(np.arange(len(sizes)), th, True)
    b2 = ctx.alloc_stats()[2]
    ctx.close()
    need_k = sum(8 * (64 * ((n + 63) // 64)) ** 2 for n in sizes)     # lower bound: every entry at its own padded size
    assert b1 - b0 < 2.6 * need_k + (8 << 20), (b0, b1, need_k)        # K only (+ class padding, + the small vectors / tables)
    assert b2 - b1 >= need_k                                           # the gradient call adds Linv (+ slab)
    np.testing.assert_allclose(nl0, nl1, rtol=1e-12)
    assert (st0 == 0).all() and (st1 == 0).all()


def test_memory_waves_same_bits_and_oracle(monkeypatch):
    sizes = SIZES
    pts, th = _cohort(sizes)
    slots = np.arange(len(sizes))
    # pinned schedule: results must not depend on how the call is cut into waves
    ref = _ctx(sizes, pts, pin=True)
    nl_r, g_r, st_r = ref.nlml_grad(slots, th, True)
    nplan_r = len(ref.last_plan())
    ref.close()
    monkeypatch.setenv("MEDGP_MEM_BUDGET_GB", "0.03")      # 32 MB of matrices per wave: N = 1100 alone (2 x 11.5 MB) nearly fills one
    cut = _ctx(sizes, pts, pin=True)
    nl_c, g_c, st_c = cut.nlml_grad(slots, th, True)
    assert len(cut.last_plan()) >= nplan_r
    mapped = cut.alloc_stats()[2]
    # outputs that need every entry's matrix afterwards cannot run in waves: refused, not wrong
    with pytest.raises(capi.MedgpError):
        cut.nlml_grad(slots, th, True, keep_factor=True)
    nl_c2, _, st_c2 = cut.nlml_grad(slots, th, False)
    cut.close()
    assert np.array_equal(nl_r, nl_c) and np.array_equal(g_r, g_c) and np.array_equal(st_r, st_c)
    np.testing.assert_allclose(nl_c2, nl_c, rtol=1e-12)
    assert mapped < 400 << 20
    # default routing in waves against the oracle
    cut2 = _ctx(sizes, pts)
    nl_d, g_d, st_d = cut2.nlml_grad(slots, th, True)
    cut2.close()
    prior = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
    for p in (0, 3, 8, 10, 15):
        m, t, y = pts[p]
        r = O.nlml_grad(7, Q, D, R, m, t, y, th[p], prior=prior, nthreads=8)
        assert st_d[p] == r["status"] == 0
        assert abs(nl_d[p] - r["nlml"]) <= 1e-10 * abs(r["nlml"])
        gs = np.abs(r["grad"]).max()
        assert np.all(np.abs(g_d[p] - r["grad"]) <= 1e-6 * np.maximum(np.abs(r["grad"]), 1e-3 * gs))


def test_growing_the_buffers_changes_no_bit():
    """A context whose buffers are replaced by larger ones between calls (small call first, no plan announced) returns the bits of a
    context that was sized once -- and the factor export works on the grown buffers."""
    sizes = SIZES[:10]
    pts, th = _cohort(sizes)
    slots = np.arange(len(sizes))

    def run(plan):
        ctx = medgp_amd.Context(7, Q, D, R)
        ctx.reserve(len(sizes), 6000, 64)
        ctx.set_patients(slots, pts)
        if plan:
            ctx.reserve_plan(sizes)
        small = ctx.nlml_grad(slots[1:3].copy(), th[1:3], True)      # small first: without a plan the next call outgrows every buffer
        full = ctx.nlml_grad(slots, th, True)
        alpha, linv, beta = ctx.get_factor(0, sizes[0])
        fac = (alpha, linv, np.float32(beta))
        calls = ctx.alloc_stats()[1]
        ctx.close()
        return small, full, fac, calls

    a = run(False)
    b = run(True)
    for x, y in zip(a[0] + a[1], b[0] + b[1]):
        assert np.array_equal(x, y)
    for x, y in zip(a[2], b[2]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert a[3] > b[3]        # the unplanned context went back to the allocator between the calls


def test_growth_under_an_asynchronous_lane_in_flight():
    """A blocking call outgrows every buffer while an asynchronous lane's kernels -- queued just before, on the same buffers -- may still
    be running: the outgrown blocks are retired, not freed, so the lane's results are those of an undisturbed context, bit for bit, and
    so are the blocking call's.  (Round 5 synchronised the device and freed at this point.)"""
    sizes = SIZES
    pts, th = _cohort(sizes)
    small = np.array([1, 4, 6, 9, 10])            # N <= 90
    allp = np.arange(len(sizes))

    ref = _ctx(sizes, pts)
    want_small = ref.nlml_grad(small, th[small], True)
    want_all = ref.nlml_grad(allp, th, True)
    ref.close()

    for rep in range(3):
        ctx = medgp_amd.Context(7, Q, D, R)
        ctx.reserve(len(sizes), 6000, 64)          # buffers on demand
        ctx.set_patients(allp, pts)
        ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
        H = ctx.H
        b = dict(th=ctx.pinned((len(small), H), np.float64), nl=ctx.pinned((len(small),), np.float64), gr=ctx.pinned((len(small), H), np.float64),
                 st=ctx.pinned((len(small),), np.int32))
        b["th"][:] = th[small]
        b["gr"][:] = -7.0
        ctx.nlml_grad_async(rep % 2, small, b["th"], True, b["nl"], b["gr"], b["st"])   # sizes the buffers for five small patients
        got_all = ctx.nlml_grad(allp, th, True)                                         # ... and this one outgrows all of them at once
        ctx.wait(rep % 2)
        s = ctx.alloc_stats()
        ctx.close()
        for x, y in zip(want_all, got_all):
            assert np.array_equal(x, y)
        assert np.array_equal(b["nl"], want_small[0]) and np.array_equal(b["gr"], want_small[1]) and np.array_equal(b["st"], want_small[2])
        assert s[2] > 0
