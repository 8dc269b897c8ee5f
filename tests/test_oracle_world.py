"""The oracle's statement of DECOMPOSED runs (oracle/orc_md.c: orc_world): bricks, per-rank bead order, ghosts from other
ranks, per-rank RanMars streams (seed + me; UCG/fix_ucgld_langevin.cpp:85, 280, UCG/fix_ucgstate.cpp:62, 117).  CPU tier:
its own consistency -- one brick is the single-rank oracle bit for bit, several bricks give the same physics.  The GPU
library's decomposed loop is compared with it bit for bit in tests/test_multi_rank.py."""
import numpy as np
import pytest

import util


def _by_tag(n, tag, arr):
    out = np.zeros((n,) + np.asarray(arr).shape[1:])
    out[np.asarray(tag) - 1] = arr
    return out


def _world(orc, beads, deck, grid, style="table_ucgld", dt=0.004, every=2, lang=(1.0, 1.0, 1.0, 48279), ust=("mc", 9127, 0.3), nve="wall"):
    op = util.oracle_pair(style, deck)
    w = orc.World(beads, grid)
    w.set_run_params(dt=dt, every=every, delay=0, check=1)
    w.attach(op, langevin=lang, nve=nve, ucgstate=ust)
    return w, op


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe", "table_ucg_bethe_density"])
def test_one_brick_is_the_single_rank_oracle_bit_for_bit(orc, pkg, style):
    dens = dict(density=(11.3, 1.5), extra11=0.05) if style.endswith("density") else {}
    deck = util.make_deck("spline", 1024, **dens)
    beads = pkg.synth.make_beads(7, seed=5)
    lang = (1.0, 1.0, 1.0, 48279) if style == "table_ucgld" else None
    w, op = _world(orc, beads, deck, [1, 1, 1], style=style, lang=lang)
    sim = util.oracle_sim(beads, op, mode=1, dt=0.004, langevin=lang, nve="wall", ucgstate=("mc", 9127, 0.3), every=2)
    assert w.setup(60) == 0 and sim.setup(60) == 0
    assert w.run(60, 30) == 0 and sim.run(60, 30) == 0
    A, B = w.rank_arrays(0, ghosts=True), sim.arrays(ghosts=True)
    assert (A["nlocal"], A["nghost"]) == (B["nlocal"], B["nghost"])
    assert np.array_equal(A["tag"], B["tag"]) and np.array_equal(A["ucgstate"], B["ucgstate"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(A[k], B[k]), k
    assert w.rank_info(0)["nrebuild"] == sim.info()["nrebuild"] >= 3
    assert w.ev()["eng_vdwl"] == sim.ev()["eng_vdwl"]


@pytest.mark.parametrize("grid", [[2, 1, 1], [2, 2, 1], [2, 2, 2]])
def test_bricks_partition_the_beads_and_reproduce_the_single_rank_physics(orc, pkg, grid):
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(10, seed=5)
    n = beads.n
    # no thermostat: the trajectory does not depend on the decomposition beyond the order of the sums
    w, op = _world(orc, beads, deck, grid, lang=None, ust=None, nve=True)
    sim = util.oracle_sim(beads, op, mode=0, dt=0.004, nve=True, every=2)
    assert w.setup(40) == 0 and sim.setup(40) == 0
    R = [w.rank_arrays(r) for r in range(w.nranks)]
    tags = np.concatenate([a["tag"] for a in R])
    assert sorted(tags.tolist()) == list(range(1, n + 1))  # every bead on exactly one rank
    m = pkg.multi
    for r, a in enumerate(R):
        lo, hi = m.sub_box(beads.boxlo, beads.boxhi, grid, r)
        assert np.all((a["x"] >= lo) & (a["x"] < hi)) and a["nghost"] > 0
    S = sim.arrays()
    for k in ("f", "ucgforce", "scores"):
        multi = _by_tag(n, tags, np.concatenate([a[k] for a in R]))
        ref = _by_tag(n, S["tag"], S[k])
        assert np.abs(multi - ref).max() <= 1e-11 * np.abs(ref).max(), k
    assert abs(w.ev()["eng_vdwl"] - sim.ev()["eng_vdwl"]) <= 1e-11 * abs(sim.ev()["eng_vdwl"])
    assert w.run(40, 40) == 0 and sim.run(40, 40) == 0
    R = [w.rank_arrays(r) for r in range(w.nranks)]
    tags = np.concatenate([a["tag"] for a in R])
    assert sorted(tags.tolist()) == list(range(1, n + 1))
    S = sim.arrays()
    d = _by_tag(n, tags, np.concatenate([a["x"] for a in R])) - _by_tag(n, S["tag"], S["x"])
    d -= np.round(d / beads.boxhi) * beads.boxhi
    assert np.abs(d).max() < 1e-9
    assert all(w.rank_info(r)["nrebuild"] == sim.info()["nrebuild"] for r in range(w.nranks))


def test_density_style_on_bricks_takes_its_ghosts_priors_from_the_owner_ranks(orc, pkg):
    """table_ucg_bethe_density decomposed: the priors (after pass 1) and CV forces (after pass 2) of ghosts come from the ranks
    that own them; forces and posteriors agree with the single-rank oracle to rounding, momentum is conserved"""
    deck = util.make_deck("spline", 1024, density=(11.3, 1.5), extra11=0.05)
    beads = pkg.synth.make_beads(10, seed=5)
    n = beads.n
    w, op = _world(orc, beads, deck, [2, 2, 1], style="table_ucg_bethe_density", dt=0.002, lang=None, ust=None, nve=True)
    sim = util.oracle_sim(beads, op, mode=1, dt=0.002, nve=True, every=2)
    assert w.setup(20) == 0 and sim.setup(20) == 0
    R = [w.rank_arrays(r) for r in range(4)]
    tags = np.concatenate([a["tag"] for a in R])
    S = sim.arrays()
    for k, tol in (("f", 1e-10), ("ucgp", 1e-12)):
        multi, ref = _by_tag(n, tags, np.concatenate([a[k] for a in R])), _by_tag(n, S["tag"], S[k])
        assert np.abs(multi - ref).max() <= tol * np.abs(ref).max(), k
    f = np.concatenate([a["f"] for a in R])
    assert np.abs(f.sum(axis=0)).max() < 1e-9 * np.abs(f).max() * n
    assert abs(w.ev()["eng_vdwl"] - sim.ev()["eng_vdwl"]) <= 1e-11 * abs(sim.ev()["eng_vdwl"])
    assert w.run(20, 20) == 0 and sim.run(20, 20) == 0
    R = [w.rank_arrays(r) for r in range(4)]
    tags = np.concatenate([a["tag"] for a in R])
    d = _by_tag(n, tags, np.concatenate([a["x"] for a in R])) - _by_tag(n, sim.arrays()["tag"], sim.arrays()["x"])
    d -= np.round(d / beads.boxhi) * beads.boxhi
    assert np.abs(d).max() < 1e-9


def test_thermostatted_decomposed_run_migrates_beads_and_is_reproducible(orc, pkg):
    """per-rank streams: the draws follow the ranks' local bead order, so the trajectory is a function of the decomposition
    -- and of nothing else (two runs agree bit for bit); beads cross the brick faces during the run"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(10, seed=5)
    runs = []
    for _ in range(2):
        w, op = _world(orc, beads, deck, [2, 1, 1])
        assert w.setup(240) == 0
        counts = [w.rank_arrays(0)["nlocal"]]
        for _ in range(4):
            assert w.run(60) == 0
            counts.append(w.rank_arrays(0)["nlocal"])
        runs.append(([w.rank_arrays(r) for r in range(2)], counts))
    assert len(set(runs[0][1])) > 1, "no migration: the test does not exercise the exchange"
    assert runs[0][1] == runs[1][1]
    for a, b in zip(runs[0][0], runs[1][0]):
        assert np.array_equal(a["tag"], b["tag"]) and np.array_equal(a["ucgstate"], b["ucgstate"])
        for k in ("x", "v", "ucgl"):
            assert util.bits_equal(a[k], b[k]), k
    lam = np.concatenate([a["ucgl"] for a in runs[0][0]])
    assert lam.min() >= 0.0 and lam.max() <= 1.0
    # a different decomposition draws differently: the thermostatted trajectories are not the same
    w1, _ = _world(orc, beads, deck, [1, 1, 1])
    assert w1.setup(240) == 0 and w1.run(240) == 0
    one = w1.rank_arrays(0)
    two_l = _by_tag(beads.n, np.concatenate([a["tag"] for a in runs[0][0]]), np.concatenate([a["ucgl"] for a in runs[0][0]]))
    assert not np.array_equal(two_l, _by_tag(beads.n, one["tag"], one["ucgl"]))
