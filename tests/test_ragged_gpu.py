"""A heavy-tailed cohort in ONE call (round 5): 300 patients whose observation counts are log-normal (median 263, eight above
N = 1400, the largest 5832), D = 24, Q = 5, R = 8, hierarchical-gamma prior -- the shape real cohorts have and the reference's job
generator schedules by size (ref: scripts/slurm_della.json:6-62, medgpc/util/run_exp_generator.py:213-260).

The library cuts the call into size classes (by 64-block count), gives every class its own leading dimension, launch geometry and
factorisation route -- the heavy tail goes to the multi-CU look-ahead schedule, the bulk to one workgroup per patient -- and runs the
classes beside each other on separate streams.  Checked here on DEFAULT routing:
  * every patient above N = 1200 against committed oracle outputs (tests/golden/ragged_cohort_large.npz, made by
    tests/golden/make_ragged_oracle.py: the CPU oracle needs minutes for them), 40 of the others against the oracle run live;
  * the plan: the tail is on the look-ahead schedule, the bulk is not;
  * results do not depend on the caller's order of the entries, nor on whether the classes share one stream (same kernels, same bits).
Tolerances as in test_parity_gpu.py (north_star: <= 1e-6 relative on log-lik and gradients).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import medgp_amd
from medgp_amd import synth
from oracle import oracle as O

NLML_RTOL = 1e-10
GRAD_RTOL = 1e-6
HERE = os.path.dirname(os.path.abspath(__file__))
SEED, P, D, Q, R = 0, 300, 24, 5, 8


def _check(p, nlml, grad, ref_nlml, ref_grad):
    assert abs(nlml - ref_nlml) <= NLML_RTOL * abs(ref_nlml), (p, nlml, ref_nlml)
    gs = np.abs(ref_grad).max()
    err = np.abs(grad - ref_grad) / np.maximum(np.abs(ref_grad), 1e-3 * gs)
    assert err.max() <= GRAD_RTOL, (p, int(err.argmax()), float(err.max()))


@pytest.fixture(scope="module")
def cohort():
    pts, th, ns = synth.ragged_cohort(SEED, P, D, 7, Q, R)
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, int(ns.max()), P)
    ctx.set_patients(np.arange(P), pts)
    ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    yield ctx, pts, th, ns
    ctx.close()


def test_ragged_cohort_vs_oracle_default_routing(cohort):
    ctx, pts, th, ns = cohort
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    plan = ctx.last_plan()
    assert sum(c for c, _, _ in plan) == P
    # the heavy tail (more than 16 blocks: the eight patients above N = 1400 and their neighbours) takes the look-ahead schedule,
    # the one- and two-block classes never do, and the classes are ordered largest first
    assert all(r == 2 for c, nb, r in plan if nb > 16), plan
    assert all(r != 2 for c, nb, r in plan if nb <= 2), plan
    assert [nb for _, nb, _ in plan] == sorted((nb for _, nb, _ in plan), reverse=True)
    assert (st == 0).all(), np.where(st != 0)
    gold = np.load(os.path.join(HERE, "golden", "ragged_cohort_large.npz"))
    assert int(gold["seed"]) == SEED and int(gold["P"]) == P
    assert sorted(gold["index"].tolist()) == [p for p in range(P) if ns[p] > int(gold["large"])]
    for k, p in enumerate(gold["index"]):
        assert int(gold["n"][k]) == ns[p] and int(gold["status"][k]) == 0
        _check(int(p), nlml[p], grad[p], float(gold["nlml"][k]), gold["grad"][k])
    prior = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
    small = [p for p in range(P) if ns[p] <= int(gold["large"])]
    for p in small[::max(1, len(small) // 40)]:
        m, t, y = pts[p]
        ref = O.nlml_grad(7, Q, D, R, m, t, y, th[p], prior=prior, nthreads=8)
        assert ref["status"] == 0
        _check(p, nlml[p], grad[p], ref["nlml"], ref["grad"])
    # the nlml-only path of the same call
    nlml0, _, st0 = ctx.nlml_grad(np.arange(P), th, False)
    assert (st0 == 0).all()
    np.testing.assert_allclose(nlml0, nlml, rtol=1e-12)


def test_ragged_cohort_order_and_streams_do_not_change_bits(cohort, monkeypatch):
    ctx, pts, th, ns = cohort
    sel = np.argsort(ns, kind="stable")[:280]          # without the 20 largest: quick
    nlml, grad, st = ctx.nlml_grad(sel, th[sel], True)
    g = np.random.Generator(np.random.Philox(key=[5, 5]))
    perm = g.permutation(len(sel))
    nlml_p, grad_p, st_p = ctx.nlml_grad(sel[perm], th[sel[perm]], True)
    assert np.array_equal(nlml_p, nlml[perm]) and np.array_equal(grad_p, grad[perm]) and np.array_equal(st_p, st[perm])
    # the classes back to back on one stream: a second context (the switch is read at creation)
    monkeypatch.setenv("MEDGP_CLASS_STREAMS", "0")
    ctx2 = medgp_amd.Context(7, Q, D, R)
    ctx2.reserve(len(sel), int(ns[sel].max()), len(sel))
    ctx2.set_patients(np.arange(len(sel)), [pts[p] for p in sel])
    ctx2.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    nlml2, grad2, st2 = ctx2.nlml_grad(np.arange(len(sel)), th[sel], True)
    ctx2.close()
    assert np.array_equal(nlml2, nlml) and np.array_equal(grad2, grad) and np.array_equal(st2, st)


def test_get_factor_after_a_ragged_call(cohort):
    """medgp_get_factor reads an entry's alpha / L^-1 out of its size class's view (leading dimension of the class, not the context's)."""
    ctx, pts, th, ns = cohort
    order = np.argsort(ns, kind="stable")
    sel = np.array([order[0], order[150], order[250], order[120], order[280]])    # five classes, caller order not sorted
    ctx.nlml_grad(sel, th[sel], True)
    for b, p in enumerate(sel):
        m, t, y = pts[p]
        alpha, linv, beta = ctx.get_factor(b, int(ns[p]))
        ref = O.nlml_grad(7, Q, D, R, m, t, y, th[p], want_alpha=True, want_linv=True)
        np.testing.assert_allclose(alpha, ref["alpha"], rtol=2e-6, atol=1e-6 * np.abs(ref["alpha"]).max())
        np.testing.assert_allclose(linv, ref["linv"], rtol=2e-6, atol=1e-6 * np.abs(ref["linv"]).max())
        assert abs(beta - ref["beta"]) <= 1e-6 * abs(ref["beta"])
        assert np.all(np.triu(linv, 1) == 0)


def test_arenas_grow_with_the_calls_when_the_reservation_is_huge():
    """medgp_reserve(max_batch x max_n^2 beyond 8 GB) allocates no per-entry matrices up front: 2048 x N = 20000 would be 13 TB.  The
    calls grow the arenas to what their size classes need, results are the oracle's, and a later, larger call grows them again."""
    Dm, Qm, Rm = 2, 2, 2
    ns = [40, 700, 130, 9, 300]
    pts = [synth.patient(5, p, Dm, n) for p, n in enumerate(ns)]
    th = np.stack([synth.theta(5, p, 7, Qm, Dm, Rm) for p in range(len(ns))])
    ctx = medgp_amd.Context(7, Qm, Dm, Rm)
    ctx.reserve(8, 20000, 2048)
    ctx.set_patients(np.arange(len(ns)), pts)
    for sel in ([0, 3], [0, 1, 2, 3, 4], [1] * 40 + [2, 4]):          # growing needs: 2 small entries, all five, 40 x N = 700
        sel = np.array(sel)
        nlml, grad, st = ctx.nlml_grad(sel, th[sel], True)
        assert np.all(st == 0)
        for b in (0, len(sel) - 1):
            p = sel[b]
            ref = O.nlml_grad(7, Qm, Dm, Rm, *pts[p], th[p])
            _check(int(p), nlml[b], grad[b], ref["nlml"], ref["grad"])
    ctx.close()


def test_route_rule_of_uniform_calls_is_the_round_4_rule(monkeypatch):
    """For a uniform call the per-class rule on the cost model reduces to the measured thresholds of round 4 (profiles/r04_route_table.txt):
    look-ahead schedule up to 7/16 #CU entries (112) of three or four blocks and up to 9/16 #CU (144) from five blocks on, never for two;
    one workgroup per entry beyond -- 4-wave shape when there are more entries than CUs or at most four blocks, else 8 waves."""
    Dm, Qm, Rm = 1, 1, 0
    ctx = medgp_amd.Context(8, Qm, Dm, Rm)
    ctx.reserve(4, 330, 300)
    for s, n in enumerate((100, 200, 330)):
        ctx.set_patient(s, None, *synth.patient(3, s, 1, n)[1:])
    th = synth.theta(3, 0, 8, Qm, 1, 0)
    for slot, blocks, cases in ((0, 2, [(8, 0), (200, 0)]), (1, 4, [(112, 2), (113, 0), (300, 0)]), (2, 6, [(144, 2), (145, 1), (257, 0)])):
        for count, route in cases:
            ctx.nlml_grad(np.full(count, slot), np.tile(th, (count, 1)), False)
            assert ctx.last_plan() == [(count, blocks, route)], (blocks, count, ctx.last_plan())
    ctx.close()


def test_one_class_switch_gives_the_same_values(cohort, monkeypatch):
    """MEDGP_NO_CLASSES=1 (the A/B switch of bench.py's `before` leg: one class, one route per call as in rounds 1-4) must still be a
    correct evaluator: same statuses, values within the parity tolerance of the default plan's."""
    ctx, pts, th, ns = cohort
    sel = np.argsort(ns, kind="stable")[100:220]
    nlml, grad, st = ctx.nlml_grad(sel, th[sel], True)
    monkeypatch.setenv("MEDGP_NO_CLASSES", "1")
    ctx2 = medgp_amd.Context(7, Q, D, R)
    ctx2.reserve(len(sel), int(ns[sel].max()), len(sel))
    ctx2.set_patients(np.arange(len(sel)), [pts[p] for p in sel])
    ctx2.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    nlml2, grad2, st2 = ctx2.nlml_grad(np.arange(len(sel)), th[sel], True)
    assert len(ctx2.last_plan()) == 1 and np.array_equal(st, st2)
    np.testing.assert_allclose(nlml2, nlml, rtol=1e-12)
    gs = np.abs(grad).max(axis=1, keepdims=True)
    assert (np.abs(grad2 - grad) / np.maximum(np.abs(grad), 1e-3 * gs)).max() <= 1e-8
    ctx2.close()


@pytest.mark.parametrize("ns", [[900, 800, 700, 600], [1000, 960, 900, 840, 800, 760, 700, 590], [2000, 1100]])
def test_ragged_classes_of_two_four_and_eight_entries(ns):
    """Classes whose entries differ in block count take an odd stride of the entry index in k_wgrad's workgroup ids and an odd x extent
    of the look-ahead grids (count | 1), so that every entry's workgroups go to all 8 XCDs instead of 8 / count of them; the padding
    column's workgroups must do nothing.  One size class of 4 / 8 / 2 entries of different sizes on default routing (look-ahead
    schedule), nlml + gradient and nlml-only, against the oracle."""
    Dm, Qm, Rm = 4, 3, 2
    pts = [synth.patient(23, p, Dm, n) for p, n in enumerate(ns)]
    th = np.stack([synth.theta(23, p, 7, Qm, Dm, Rm) for p in range(len(ns))])
    ctx = medgp_amd.Context(7, Qm, Dm, Rm)
    ctx.reserve(len(ns), max(ns), len(ns))
    ctx.set_patients(np.arange(len(ns)), pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(len(ns)), th, True)
    plan = ctx.last_plan()
    assert any(cnt == len(ns) and r == 2 for cnt, _, r in plan) or len(ns) == 2, plan
    nlml0, _, st0 = ctx.nlml_grad(np.arange(len(ns)), th, False)
    assert np.all(st == 0) and np.all(st0 == 0)
    for p in range(len(ns)):
        ref = O.nlml_grad(7, Qm, Dm, Rm, *pts[p], th[p], nthreads=4)
        _check(p, nlml[p], grad[p], ref["nlml"], ref["grad"])
        assert abs(nlml0[p] - ref["nlml"]) <= NLML_RTOL * abs(ref["nlml"])
    ctx.close()
