"""GPU tests of the host-overlap surface added in round 3: batched prior upload (medgp_set_priors), the two-lane asynchronous
operator (medgp_nlml_grad_async / medgp_wait on pinned memory), slot tables that change on every call without a device wait,
and the argument checks of the packed uploads.  Reference callers: util/c_optimizer_scg.cpp:65,120,221 (one objective call per
line-search point), util/c_optimizer_varEM.cpp:98-162 (prior parameters change once per outer iteration)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import medgp_amd
from medgp_amd import capi, synth
from oracle import oracle as O


def _ctx(P=12, D=3, N=70, Q=3, R=2, max_batch=None):
    pts, th = synth.cohort(77, P, D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, N, max_batch or P)
    ctx.set_patients(np.arange(P), pts)
    return ctx, pts, th


def test_set_priors_batched_equals_per_slot():
    P, D, Q, R = 12, 3, 3, 2
    ctx, pts, th = _ctx(P, D, 70, Q, R)
    H = ctx.H
    rng = np.random.default_rng(5)
    f, ty, ex, p0, p1 = synth.hier_gamma_prior(Q, D, R, 0.01)
    rows = []
    for s in range(P):
        q1 = p1.copy()
        q1[D:D + Q * D * R] = rng.uniform(0.2, 3.0, Q * D * R).astype(np.float32)   # per-patient psi, as varEM produces them
        t2 = ty.copy()
        t2[D + rng.integers(0, Q * D * R, 3)] = 0                                    # a few clamped entries
        rows.append((f, t2, ex, p0, q1))
    for s in range(P):
        ctx.set_prior(s, *rows[s])
    ref = ctx.nlml_grad(np.arange(P), th, True)
    ctx.set_prior(-1)                                   # remove everywhere (one broadcast row)
    none = ctx.nlml_grad(np.arange(P), th, True)
    assert not np.array_equal(none[0], ref[0])
    order = rng.permutation(P)
    ctx.set_priors(order, *[np.stack([rows[s][k] for s in order]) for k in range(5)])
    got = ctx.nlml_grad(np.arange(P), th, True)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    ctx.set_priors(np.array([3, 4]))                    # flag == NULL: prior off for those two
    part = ctx.nlml_grad(np.arange(P), th, True)
    assert np.array_equal(part[0][[3, 4]], none[0][[3, 4]]) and np.array_equal(part[0][[0, 1, 2]], ref[0][[0, 1, 2]])
    ctx.set_prior(-1, f, ty, ex, p0, p1)                # one row for every slot
    allp = ctx.nlml_grad(np.arange(P), th, True)
    for s in range(P):
        ctx.set_prior(s, f, ty, ex, p0, p1)
    allq = ctx.nlml_grad(np.arange(P), th, True)
    assert np.array_equal(allp[0], allq[0]) and np.array_equal(allp[1], allq[1])
    with pytest.raises(capi.MedgpError):
        ctx.set_priors(np.array([0, P]), *[np.stack([rows[0][k]] * 2) for k in range(5)])
    bad = rows[0][1].copy()
    bad[0] = 3
    with pytest.raises(capi.MedgpError):
        ctx.set_priors(np.array([0]), rows[0][0][None], bad[None], rows[0][2][None], rows[0][3][None], rows[0][4][None])
    ctx.close()


def test_async_lanes_equal_synchronous_calls():
    P = 12
    ctx, pts, th = _ctx(P)
    H = ctx.H
    ref = ctx.nlml_grad(np.arange(P), th, True)
    ref0 = ctx.nlml_grad(np.arange(P), th, False)
    halves = [np.arange(0, 7), np.arange(7, P)]
    bufs = []
    for lane, idx in enumerate(halves):
        n = idx.size
        b = dict(th=ctx.pinned((n, H), np.float64), nl=ctx.pinned((n,), np.float64), gr=ctx.pinned((n, H), np.float64),
                 st=ctx.pinned((n,), np.int32))
        b["th"][:] = th[idx]
        b["gr"][:] = -7.0
        bufs.append(b)
    for rep in range(3):      # both lanes in flight, different slot tables back to back (no device wait in between)
        for lane, idx in enumerate(halves):
            ctx.nlml_grad_async(lane, idx, bufs[lane]["th"], True, bufs[lane]["nl"], bufs[lane]["gr"], bufs[lane]["st"])
        with pytest.raises(capi.MedgpError):          # a lane holds one call at a time
            ctx.nlml_grad_async(0, halves[0], bufs[0]["th"], True, bufs[0]["nl"], bufs[0]["gr"], bufs[0]["st"])
        for lane, idx in enumerate(halves):
            ctx.wait(lane)
            assert np.array_equal(bufs[lane]["nl"], ref[0][idx]) and np.array_equal(bufs[lane]["gr"], ref[1][idx])
            assert np.array_equal(bufs[lane]["st"], ref[2][idx])
    # nlml-only on a lane, pageable memory (allowed: the copies are then synchronous)
    nl = np.empty(5)
    st = np.empty(5, np.int32)
    ctx.nlml_grad_async(1, np.arange(5), np.ascontiguousarray(th[:5]), False, nl, None, st)
    ctx.wait(1)
    assert np.array_equal(nl, ref0[0][:5])
    ctx.wait(0)   # nothing pending: no-op
    ctx.close()


def test_slot_table_changes_every_call_without_sync():
    """The lock-step optimiser shrinks / reorders its active set on most steps: many calls with different slot lists queued
    back to back must each see their own table (it travels through a pinned ring, stream ordered)."""
    import torch
    P = 12
    ctx, pts, th = _ctx(P)
    H = ctx.H
    ref = ctx.nlml_grad(np.arange(P), th, True)
    dev = torch.device("cuda", 0)
    th_d = torch.from_numpy(th).to(dev)
    rng = np.random.default_rng(9)
    outs = []
    for it in range(40):
        k = int(rng.integers(1, P + 1))
        idx = rng.permutation(P)[:k]
        t_in = th_d[torch.from_numpy(idx).to(dev)].contiguous()
        nl = torch.empty(k, dtype=torch.float64, device=dev)
        gr = torch.empty((k, H), dtype=torch.float64, device=dev)
        st = torch.empty(k, dtype=torch.int32, device=dev)
        ctx.nlml_grad_device(idx, t_in.data_ptr(), 1, nl.data_ptr(), gr.data_ptr(), st.data_ptr())
        outs.append((idx, t_in, nl, gr, st))
    ctx.synchronize()
    torch.cuda.synchronize()
    for idx, _, nl, gr, st in outs:
        assert np.array_equal(nl.cpu().numpy(), ref[0][idx]) and np.array_equal(gr.cpu().numpy(), ref[1][idx])
    ctx.close()


def test_packed_upload_rejects_duplicate_slots():
    ctx, pts, th = _ctx(4)
    with pytest.raises(capi.MedgpError) as e:
        ctx.set_patients(np.array([1, 2, 1]), [pts[0], pts[1], pts[2]])
    assert "twice" in str(e.value)
    nl = ctx.nlml_grad(np.arange(4), th, False)[0]          # nothing was touched by the rejected call
    ctx.set_patients(np.array([1, 2]), [pts[1], pts[2]])
    assert np.array_equal(ctx.nlml_grad(np.arange(4), th, False)[0], nl)
    ctx.close()


def test_kde_grid_with_non_finite_point_is_reported():
    rng = np.random.default_rng(3)
    x = [rng.normal(size=200), rng.normal(size=150)]
    grid = [np.linspace(-3, 3, 50), np.linspace(-3, 3, 40)]
    ok = capi.kde_mode(x, False, 0, full=True, test=grid)
    assert np.all(ok[2] == 0)
    grid[1] = grid[1].copy()
    grid[1][7] = np.nan
    mode, bw, st, _ = capi.kde_mode(x, False, 0, full=True, test=grid)
    assert st[0] == 0 and mode[0] == ok[0][0] and st[1] == -1 and np.isnan(mode[1])


def test_pinned_ring_wraps_without_corrupting_queued_uploads():
    """The small-upload ring (slot tables, prior rows, theta of small calls) is two pinned buffers of 1 MB: several thousand
    queued uploads force it to wrap many times while earlier copies may still be in flight.  Alternating prior sets and slot
    lists must always be evaluated with the values of THEIR call."""
    import torch
    P, D, Q, R = 12, 3, 3, 2
    ctx, pts, th = _ctx(P, D, 70, Q, R)
    H = ctx.H
    f, ty, ex, p0, p1 = synth.hier_gamma_prior(Q, D, R, 0.01)
    sets = []
    for v in (0.5, 2.0):
        q1 = p1.copy()
        q1[D:D + Q * D * R] = np.float32(v)
        sets.append([np.stack([a] * P) for a in (f, ty, ex, p0, q1)])
    ref = []
    for k in range(2):
        ctx.set_priors(np.arange(P), *sets[k])
        ref.append(ctx.nlml_grad(np.arange(P), th, False)[0])
    assert not np.array_equal(ref[0], ref[1])
    dev = torch.device("cuda", 0)
    th_d = torch.from_numpy(th).to(dev)
    outs = []
    row_bytes = P * H * 12
    ncalls = int(3 * (2 << 20) / row_bytes) + 8          # > three full revolutions of the ring
    for it in range(ncalls):
        k = it & 1
        ctx.set_priors(np.arange(P), *sets[k])            # queued, no device wait
        if it % 16 == 0:
            nl = torch.empty(P, dtype=torch.float64, device=dev)
            ctx.nlml_grad_device(np.arange(P), th_d.data_ptr(), 0, nl.data_ptr(), 0, 0)
            outs.append((k, nl))
    ctx.synchronize()
    torch.cuda.synchronize()
    assert len(outs) > 20
    for k, nl in outs:
        assert np.array_equal(nl.cpu().numpy(), ref[k])
    ctx.close()


def test_profile_only_one_kernel_brackets_just_that_kernel():
    """medgp_profile_enable(ctx, 2 + k) (bench.py's timed region): events around the launches of ONE kernel; results untouched."""
    P, D, N, Q, R = 6, 3, 130, 2, 2
    pts, th = synth.cohort(5, P, D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, N, P)
    ctx.set_patients(np.arange(P), pts)
    ref = ctx.nlml_grad(np.arange(P), th, True)
    ctx.profile_reset()
    ctx.profile_enable(True, only="k_wgrad")
    for _ in range(3):
        got = ctx.nlml_grad(np.arange(P), th, True)
    prof = {k: v for k, v in ctx.profile_read().items() if v[1] > 0}
    ctx.profile_enable(False)
    assert list(prof) == ["k_wgrad"] and prof["k_wgrad"][1] == 3 and prof["k_wgrad"][0] > 0.0
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    ctx.profile_reset()
    ctx.profile_enable(True)
    ctx.nlml_grad(np.arange(P), th, True)
    full = {k for k, v in ctx.profile_read().items() if v[1] > 0}
    ctx.profile_enable(False)
    assert {"k_prep", "k_assemble", "k_wgrad", "k_epilogue"} <= full


def test_factor_batch_equals_single_factor_calls_and_pin_route_fixes_the_bits():
    """medgp_factor_batch == one medgp_factor per entry (same L, z, status, ragged sizes, caller order kept), and with the route
    pinned (medgp_pin_route) a patient's nlml / gradient / factor do not depend on the batch it is evaluated in: alone (where the
    library would otherwise take the multi-CU look-ahead schedule) vs among 300 batch-mates (where it would take k_cholinv<4,4>)."""
    import medgp_amd
    from medgp_amd import synth
    D, Q, R = 3, 2, 2
    ns = [150, 64, 333, 7, 200]
    pts = [synth.patient(41, s, D, n, interleave=(s % 2 == 1)) for s, n in enumerate(ns)]
    th = np.stack([synth.theta(41, s, 7, Q, D, R) for s in range(len(ns))])
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(304, max(ns), 304)
    ctx.pin_route(True)
    ctx.set_patients(np.arange(len(ns)), pts)
    got, st = ctx.factor_batch(np.arange(len(ns)), th, ns)
    assert np.all(st == 0)
    for s, n in enumerate(ns):
        L1, z1, st1 = ctx.factor(s, th[s], n)
        assert st1 == 0 and np.array_equal(L1, got[s][0]) and np.array_equal(z1, got[s][1])
        assert np.all(np.triu(got[s][0], 1) == 0.0)
        m, t, y = pts[s]
        ref = O.nlml_grad(7, Q, D, R, m, t, y, th[s], want_linv=True)      # caller order: L L^T = K + noise of the caller's ordering
        Linv = ref["linv"]
        assert np.max(np.abs(Linv @ got[s][0] - np.eye(n))) < 1e-8
    # the pinned route: one patient alone vs the same patient as entry 299 of a batch of 300
    big = synth.patient(43, 0, D, 300)
    thb = synth.theta(43, 0, 7, Q, D, R)
    ctx.set_patients(np.arange(300), [big if s == 299 else pts[s % 5] for s in range(300)])
    tha = np.stack([thb if s == 299 else th[s % 5] for s in range(300)])
    nl_b, g_b, st_b = ctx.nlml_grad(np.arange(300), tha, True)
    nl_1, g_1, st_1 = ctx.nlml_grad([299], thb[None, :], True)
    assert st_b[299] == 0 and st_1[0] == 0
    assert nl_b[299] == nl_1[0] and np.array_equal(g_b[299], g_1[0])
    refb = O.nlml_grad(7, Q, D, R, *big, thb)
    assert abs(nl_1[0] - refb["nlml"]) <= 1e-10 * abs(refb["nlml"])
    ctx.close()


def test_set_priors_rejects_a_duplicate_slot():
    """Two rows for one slot in a single medgp_set_priors would race in the scatter kernel (advisor finding, round 3): rejected with
    MEDGP_ERR_ARG, nothing is changed."""
    P, D, Q, R = 4, 3, 3, 2
    ctx, pts, th = _ctx(P, D, 70, Q, R)
    f, ty, ex, p0, p1 = synth.hier_gamma_prior(Q, D, R, 0.01)
    nl0, g0, _ = ctx.nlml_grad(np.arange(P), th, True)
    with pytest.raises(capi.MedgpError, match="twice"):
        ctx.set_priors([1, 2, 1], np.stack([f] * 3), np.stack([ty] * 3), np.stack([ex] * 3), np.stack([p0] * 3), np.stack([p1] * 3))
    nl1, g1, _ = ctx.nlml_grad(np.arange(P), th, True)
    assert np.array_equal(nl0, nl1) and np.array_equal(g0, g1)
    ctx.close()


def test_screen_equals_the_operator_and_the_oracle():
    """medgp_screen (HOT LOOP A, ref: main_one_train.cpp:228-253): one block of hyper vectors evaluated on several patients.  Bit for
    bit what medgp_nlml_grad(flag_grad = 0) gives for the same (patient, vector) entries in one call, the oracle's values to
    tolerance, statuses included (a patient with n <= 2 fails, ref: util/c_objective_one.cpp:51); chunked calls (more entries than
    max_batch) and calls while an asynchronous lane is in flight give the same bits."""
    D, Q, R = 3, 2, 2
    ns = [70, 2, 130, 33, 260]
    pts = [synth.patient(61, p, D, n) for p, n in enumerate(ns)]
    ninit = 7
    th = np.stack([synth.theta(61, 100 + k, 7, Q, D, R) for k in range(ninit)])
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(len(ns), max(ns), len(ns) * ninit)
    ctx.set_patients(np.arange(len(ns)), pts)
    slots = np.array([4, 0, 1, 3, 2])
    nl, st = ctx.screen(slots, th)
    rep = np.repeat(slots, ninit)
    nl_ref, _, st_ref = ctx.nlml_grad(rep, np.tile(th, (len(slots), 1)), False)
    assert np.array_equal(st, st_ref.reshape(len(slots), ninit))
    assert np.array_equal(nl, nl_ref.reshape(len(slots), ninit), equal_nan=True)
    for a, s in enumerate(slots):
        for k in range(ninit):
            ref = O.nlml_grad(7, Q, D, R, *pts[s], th[k], flag_grad=False)
            assert st[a, k] == ref["status"]
            if ref["status"] == 0:
                assert abs(nl[a, k] - ref["nlml"]) <= 1e-10 * abs(ref["nlml"])
            else:
                assert np.isnan(nl[a, k])
    # chunked (max_batch 8 < 35 entries), pinned route on both sides: the bits do not depend on how the entries fall into calls
    ctx.pin_route(True)
    nl_p, st_p = ctx.screen(slots, th)
    ctx2 = medgp_amd.Context(7, Q, D, R)
    ctx2.reserve(len(ns), max(ns), 8)
    ctx2.pin_route(True)
    ctx2.set_patients(np.arange(len(ns)), pts)
    nl2, st2 = ctx2.screen(slots, th)
    assert np.array_equal(st2, st_p) and np.array_equal(nl2, nl_p, equal_nan=True)
    # ... nor on an asynchronous lane being in flight (the screening shares lane 0's result staging)
    H = th.shape[1]
    lane_th = ctx.pinned((3, H), np.float64); lane_th[:] = th[:3]
    lane_nl = ctx.pinned((3,), np.float64)
    lane_gr = ctx.pinned((3, H), np.float64)
    lane_st = ctx.pinned((3,), np.int32)
    ctx.nlml_grad_async(0, np.array([0, 2, 4]), lane_th, True, lane_nl, lane_gr, lane_st)
    nl3, st3 = ctx.screen(slots, th)
    ctx.wait(0)
    assert np.array_equal(nl3, nl_p, equal_nan=True) and np.array_equal(st3, st_p)
    want_nl, want_gr, want_st = ctx.nlml_grad(np.array([0, 2, 4]), th[:3], True)
    assert np.array_equal(lane_nl, want_nl) and np.array_equal(lane_gr, want_gr) and np.array_equal(lane_st, want_st)
    ctx.close(); ctx2.close()
