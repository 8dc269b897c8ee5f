"""Option pair_once ("own-block pairs once", csrc/ucg_pair.hip): pairs of two beads of one 512-bead workgroup block are
evaluated by ONE of the two lanes, which adds the partner's share to LDS accumulators in fixed point (order-free, so
still bit-reproducible).  Against the oracle's statement of that order (orc_pair_set_once): forces / ucgforce / scores
bit for bit, the reference-order loop to 1e-11, whole trajectories bit for bit through rebuilds and the fused epilogue."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

B = 512


def _gpu(ctx, beads, deck, dt=0.002, every=1):
    ctx.set_units(1.0, 1.0, 1.0, dt)
    ctx.set_option("pair_once", 1)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=every, delay=0, check=1)
    gp = util.gpu_pair(ctx, "table_ucgld", deck)
    assert gp.gather_slots == 2
    return gp


def _oracle(beads, deck, mode=1, **kw):
    op = util.oracle_pair("table_ucgld", deck, slots=2)
    op.set_once(B)
    return op, util.oracle_sim(beads, op, mode=mode, **kw)


@pytest.mark.parametrize("tabstyle,tablength", [("spline", 1024), ("linear", 1024), ("lookup", 512)])
def test_once_rows_and_forces_bitwise(fresh_ctx, pkg, orc, tabstyle, tablength):
    deck = util.make_deck(tabstyle, tablength)
    beads = pkg.synth.make_beads(13, seed=11)  # 2197 beads: five workgroup blocks, the last one partial
    ctx = fresh_ctx
    gp = _gpu(ctx, beads, deck)
    ctx.neigh_rebuild()
    info = ctx.md_info()
    assert info["once_beads"] == B and 0 < info["once_maxin"] < 100
    # the rows: every pair of two owned beads of one block is in exactly one of the two rows, every other pair in both
    il, nn, fi, ne = ctx.neigh_download()
    n = len(nn)
    row = np.repeat(np.arange(n), nn)
    m = ne & 0x1FFFFFFF
    own = (m < n) & (row // B == m // B)
    key = np.minimum(row[own], m[own]).astype(np.int64) * n + np.maximum(row[own], m[own])
    assert len(np.unique(key)) == len(key) > 0
    keep = (row[own] < m[own]) != (((row[own] + m[own]) & 1) != 0)
    assert keep.all()
    oth = ~own & (m < n)
    k2 = row[oth].astype(np.int64) * n + m[oth]
    assert np.array_equal(np.sort(k2), np.sort(m[oth].astype(np.int64) * n + row[oth]))  # symmetric
    assert info["list_entries"] == len(m) + int(own.sum())
    # forces
    op, sim = _oracle(beads, deck)
    sim.rebuild()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    out = gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    assert np.array_equal(G["tag"], O["tag"])
    for k in ("f", "ucgforce", "scores"):
        assert util.bits_equal(G[k], O[k]), k
    ev = sim.ev()
    assert abs(out[0] - ev["eng_vdwl"]) <= 1e-12 * abs(ev["eng_vdwl"])
    assert np.allclose(out[1], ev["virial"], rtol=1e-11, atol=1e-9)
    # the reference's own loop (half list, scatter, doubles throughout): the fixed-point terms are 2^-40 apart
    op0 = util.oracle_pair("table_ucgld", deck)
    sim0 = util.oracle_sim(beads, op0, mode=0)
    sim0.rebuild()
    assert sim0.compute_forces(1, 1) == 0
    R = sim0.arrays()
    for k in ("f", "ucgforce", "scores"):
        assert np.max(np.abs(R[k] - G[k])) <= 1e-11 * np.max(np.abs(R[k])), k
    assert abs(np.sum(G["f"])) < 1e-8  # Newton's third law survives the fixed-point images (they are antisymmetric)


@pytest.mark.parametrize("integrator,ucgstate", [("wall", "ld"), (True, ("mc", 9127, 0.3))])
def test_once_trajectory_bitwise(fresh_ctx, pkg, orc, integrator, ucgstate):
    """langevin + ucgstate through the resident loop (per-bead hooks in the ONCE kernel's epilogue), several rebuilds"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(12, seed=3)
    steps, dt = 60, 0.004
    op, sim = _oracle(beads, deck, dt=dt, langevin=(1.0, 1.0, 1.0, 48279), nve=integrator, ucgstate=ucgstate, every=2)
    assert sim.setup(steps) == 0
    assert sim.run(steps, 20) == 0
    ctx = fresh_ctx
    gp = _gpu(ctx, beads, deck, dt=dt, every=2)
    ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279)
    if ucgstate == "ld":
        ctx.fix_ucgstate("ld")
    else:
        ctx.fix_ucgstate(*ucgstate)
    if integrator == "wall":
        ctx.fix_nve_ucgld_wall_hard(False, 0.1)
    ctx.md_attach(gp, nve=integrator, langevin=True, ucgstate=True)
    ctx.md_setup(steps)
    ctx.md_run(steps, 20)
    gp.check_errors()
    assert ctx.md_info()["once_beads"] == B
    assert ctx.md_info()["nrebuild"] == sim.info()["nrebuild"] >= 4
    G, O = ctx.atoms_download(), sim.arrays()
    assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["ucgstate"], O["ucgstate"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(G[k], O[k]), k
    assert abs(ctx.md_thermo()["eng_vdwl"] - sim.ev()["eng_vdwl"]) <= 1e-12 * abs(sim.ev()["eng_vdwl"])


def test_once_is_off_where_it_does_not_apply(fresh_ctx, pkg):
    """tables that leave no room for the accumulators in LDS, and other styles, keep whole rows and the plain kernels"""
    ctx = fresh_ctx
    beads = pkg.synth.make_beads(10, seed=2)
    deck = util.make_deck("spline", 1024, extra_keywords=("method", "bethe", "pseudo", "yes", "prior", "ucgl"))
    ctx.set_option("pair_once", 1)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    gp = util.gpu_pair(ctx, "table_ucg_bethe", deck)
    ctx.neigh_rebuild()
    assert ctx.md_info()["once_beads"] == 0 and gp.gather_slots == 1
    gp.compute(0, 0)
    gp.check_errors()
