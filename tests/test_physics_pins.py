"""Reference-independent physics pins (the reference ships no vectors, so these are the evidence that the oracle and the
kernels did not misread the same line): forces must be the negative gradient of what the styles call the energy.

* table_ucgld: E_pair = sum_pairs sum_ab w_ab(lambda) u_ab(r) and fpair = sum_ab w_ab f_ab (UCG/pair_table_ucgld.cpp:507-509):
  an exact derivative, checked by central finite differences of the reported E_pair.
* table_ucg_bethe `method mf`: p_ab = p_i p_j does not depend on x, so the same holds.
* table_ucg_bethe `method bethe`: the reported energy sum p_ab u_ab is NOT what the force derives from; the pair
  probabilities solve dF/dp11 = 0 for F = sum_ab p_ab (u_ab + kT ln p_ab) (the "variational principle" of the comment at
  UCG/pair_table_ucgld.cpp:510-511 / the closure UCG/pair_table_ucg_bethe.cpp:544-581), so f = -dF/dx with dp/dx dropping
  out.  F is rebuilt here in numpy from the tables (Pair::single) and an independent solution of the closure.
* Pair::single (UCG/pair_table_ucgld.cpp:1474-1520) against the kernels for lambda in {0, 1}.
The same checks run on the oracle (CPU tier) and on the HIP kernels (GPU tier)."""
import numpy as np
import pytest

import util

H = 1.0e-5


def _oracle_forces(pkg, beads, style, deck, mode=1):
    op = util.oracle_pair(style, deck)
    sim = util.oracle_sim(beads, op, mode=mode)
    sim.rebuild()
    assert sim.compute_forces(1, 1) == 0
    A = sim.arrays()
    o = np.argsort(A["tag"])
    return A["f"][o], sim.ev()["eng_vdwl"]


def _gpu_forces(ctx, pkg, beads, style, deck):
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    ctx.neigh_rebuild()
    gp = util.gpu_pair(ctx, style, deck)
    e, _ = gp.compute(1, 1)
    gp.check_errors()
    A = ctx.atoms_download()
    o = np.argsort(A["tag"])
    gp.close()
    return A["f"][o], e


def _copy(pkg, b):
    c = pkg.synth.make_beads(5, seed=1)
    for k in ("x", "v", "type", "tag", "mask", "ucgstate", "ucgl", "ucgvl", "ucgml", "ucgp"):
        setattr(c, k, np.array(getattr(b, k), copy=True))
    c.n, c.boxlo, c.boxhi, c.mass, c.ntypes = b.n, b.boxlo.copy(), b.boxhi.copy(), b.mass.copy(), b.ntypes
    return c


def _min_gap_to_cutoff(beads, i):
    d = beads.x - beads.x[i]
    d -= np.round(d / beads.boxhi) * beads.boxhi
    r = np.sqrt((d * d).sum(axis=1))
    r[i] = 0.0
    return np.min(np.abs(r - 2.5))


def _fd_energy_gradient(pkg, beads, forces_fn, picks):
    f0, _ = forces_fn(beads)
    worst = 0.0
    for i, d in picks:
        if _min_gap_to_cutoff(beads, i) < 50 * H:  # the tables are cut, not shifted: E jumps when a pair crosses rc
            continue
        bp, bm = _copy(pkg, beads), _copy(pkg, beads)
        bp.x[i, d] += H
        bm.x[i, d] -= H
        _, ep = forces_fn(bp)
        _, em = forces_fn(bm)
        g = -(ep - em) / (2 * H)
        worst = max(worst, abs(g - f0[i, d]) / max(1.0, abs(f0[i, d])))
    return worst


PICKS = [(0, 0), (17, 1), (63, 2), (101, 0), (124, 1)]


def _beads(pkg, seed=31):
    b = pkg.synth.make_beads(5, seed=seed)
    b.ucgp = np.clip(np.random.default_rng(seed).uniform(size=b.n), 0.05, 0.95)
    return b


@pytest.mark.parametrize("style,extra", [("table_ucgld", ()), ("table_ucg_bethe", ("method", "mf", "prior", "ucgl"))])
def test_oracle_force_is_minus_gradient_of_reported_energy(pkg, orc, style, extra):
    deck = util.make_deck("spline", 4096, extra_keywords=extra, n_file=8000)  # fine tables: f and e are splined separately
    b = _beads(pkg)
    for mode in (0, 1):  # the reference's half-list loop and the canonical gather
        worst = _fd_energy_gradient(pkg, b, lambda bb: _oracle_forces(pkg, bb, style, deck, mode), PICKS)
        assert worst < 1e-6, (style, mode, worst)


@pytest.mark.gpu
@pytest.mark.parametrize("style,extra", [("table_ucgld", ()), ("table_ucg_bethe", ("method", "mf", "prior", "ucgl"))])
def test_gpu_force_is_minus_gradient_of_reported_energy(pkg, style, extra):
    deck = util.make_deck("spline", 4096, extra_keywords=extra, n_file=8000)  # fine tables: f and e are splined separately
    b = _beads(pkg)

    def forces(bb):
        ctx = pkg.capi.Context(0)
        try:
            return _gpu_forces(ctx, pkg, bb, style, deck)
        finally:
            ctx.close()

    assert _fd_energy_gradient(pkg, b, forces, PICKS) < 1e-6


# ---- Bethe closure: F = sum_pairs sum_ab p_ab (u_ab + kT ln p_ab), p from an independent solution of the closure

def _bethe_free_energy_of_bead(pkg, hp, b, i, kT=1.0):
    """the terms of F that involve bead i (all that changes when i moves); priors as the style takes them after the first
    call: the pair's "i" (lower tag) from ucgl, its "j" from ucgp (UCG/pair_table_ucg_bethe.cpp:199-205, :247-253)"""
    d = b.x - b.x[i]
    d -= np.round(d / b.boxhi) * b.boxhi
    rsq = (d * d).sum(axis=1)
    F = 0.0
    for m in np.flatnonzero((rsq < 6.25) & (np.arange(b.n) != i)):
        lo, hi = (i, m) if b.tag[i] <= b.tag[m] else (m, i)
        pi1, pj1 = b.ucgl[lo], b.ucgp[hi]
        u = np.array([[hp.single(1 + a, 1 + c, rsq[m])[0] for c in (0, 1)] for a in (0, 1)])
        J = u[1, 1] + u[0, 0] - u[0, 1] - u[1, 0]
        # stationarity: p11 p00 = exp(-J/kT) p10 p01 -- solved by bisection on p11, not by the style's formula
        lo11, hi11 = max(0.0, pi1 + pj1 - 1.0), min(pi1, pj1)
        g = lambda p: np.log(p) + np.log(1 + p - pi1 - pj1) - np.log(pi1 - p) - np.log(pj1 - p) + J / kT  # noqa: E731
        a_, b_ = lo11 + 1e-15, hi11 - 1e-15
        for _ in range(200):
            mid = 0.5 * (a_ + b_)
            if g(mid) > 0:
                b_ = mid
            else:
                a_ = mid
        p11 = 0.5 * (a_ + b_)
        p = np.array([[1 + p11 - pi1 - pj1, pj1 - p11], [pi1 - p11, p11]])
        F += float(np.sum(p * (u + kT * np.log(p))))
    return F


def _bethe_case(pkg):
    deck = util.make_deck("spline", 4096, extra_keywords=("method", "bethe", "pseudo", "yes", "prior", "ucgl"), n_file=8000)
    b = _beads(pkg, seed=41)
    b.ucgl = np.clip(b.ucgl, 0.05, 0.95)
    hp = pkg.capi.Pair(None, "table_ucg_bethe")  # host-only: Pair::single as the table evaluator
    hp.settings(deck.pair_style_args())
    hp.coeff(deck.pair_coeff_args())
    hp.init(2, 1.0)
    return deck, b, hp


def _bethe_fd(pkg, b, hp, f0):
    worst = 0.0
    for i, d in PICKS:
        if _min_gap_to_cutoff(b, i) < 50 * H:
            continue
        bp, bm = _copy(pkg, b), _copy(pkg, b)
        bp.x[i, d] += H
        bm.x[i, d] -= H
        g = -(_bethe_free_energy_of_bead(pkg, hp, bp, i) - _bethe_free_energy_of_bead(pkg, hp, bm, i)) / (2 * H)
        worst = max(worst, abs(g - f0[i, d]) / max(1.0, abs(f0[i, d])))
    return worst


def test_oracle_bethe_force_is_minus_gradient_of_the_bethe_free_energy(pkg, orc):
    deck, b, hp = _bethe_case(pkg)
    f0, _ = _oracle_forces(pkg, b, "table_ucg_bethe", deck, mode=0)
    assert _bethe_fd(pkg, b, hp, f0) < 1e-6


@pytest.mark.gpu
def test_gpu_bethe_force_is_minus_gradient_of_the_bethe_free_energy(fresh_ctx, pkg):
    deck, b, hp = _bethe_case(pkg)
    f0, _ = _gpu_forces(fresh_ctx, pkg, b, "table_ucg_bethe", deck)
    assert _bethe_fd(pkg, b, hp, f0) < 1e-6


# ---- Pair::single against the neighbour loop for lambda in {0, 1}

def _single_sum(pkg, b, hp):
    f = np.zeros((b.n, 3))
    for i in range(b.n):
        d = b.x[i] - b.x
        d -= np.round(d / b.boxhi) * b.boxhi
        rsq = (d * d).sum(axis=1)
        for m in np.flatnonzero((rsq < 6.25) & (np.arange(b.n) != i)):
            # lambda = 0 / 1 puts the whole weight on the table of (formal type of i's lambda, formal type of m's)
            _, fpair = hp.single(1 + int(b.ucgl[i]), 1 + int(b.ucgl[m]), rsq[m])
            f[i] += d[m] * fpair
    return f


def _single_case(pkg):
    deck = util.make_deck("spline", 1024)
    b = pkg.synth.make_beads(5, seed=13)
    b.ucgl = (np.random.default_rng(5).uniform(size=b.n) < 0.5).astype(float)
    hp = pkg.capi.Pair(None, "table_ucgld")
    hp.settings(deck.pair_style_args())
    hp.coeff(deck.pair_coeff_args())
    hp.init(2, 1.0)
    return deck, b, hp


def test_oracle_lambda_limits_equal_pair_single(pkg, orc):
    deck, b, hp = _single_case(pkg)
    f0, _ = _oracle_forces(pkg, b, "table_ucgld", deck, mode=0)
    fs = _single_sum(pkg, b, hp)
    assert np.max(np.abs(f0 - fs)) <= 1e-11 * np.max(np.abs(fs))


@pytest.mark.gpu
def test_gpu_lambda_limits_equal_pair_single(fresh_ctx, pkg):
    deck, b, hp = _single_case(pkg)
    f0, _ = _gpu_forces(fresh_ctx, pkg, b, "table_ucgld", deck)
    fs = _single_sum(pkg, b, hp)
    assert np.max(np.abs(f0 - fs)) <= 1e-11 * np.max(np.abs(fs))
