"""Option pair_vrow (csrc/ucg_pair_vrow.hip): table_ucgld / table_ucg_bethe on balanced virtual rows -- the pairs of two
beads of one 512-bead workgroup block evaluated once, a bead's terms summed as order-free 64-bit integer images (fixed
sums, include/ucg_hip.h: ucg_pair_sum_fixed).  Off by default (slower than the full-row gather kernels at 1 M beads:
DESIGN.md 4.1); these tests keep it honest against the oracle's statement of the fixed sums (orc_pair_set_sum_fixed):
forces / ucgforce / scores and whole trajectories bit for bit, on uploaded and device-built lists, ragged inputs, several
actual types, the interior / boundary launches, and decomposed runs (tests/test_multi_rank.py, vrow = 1)."""
import numpy as np
import pytest

import util
from test_gpu_edges import _case, same

pytestmark = pytest.mark.gpu

REL_REFORDER = 1e-11  # fixed sums vs the reference's half-list order: rounding at 2^-38 of the tables' reference magnitudes


def _setup_gpu(ctx, beads, dt, every):
    ctx.set_units(1.0, 1.0, 1.0, dt)
    ctx.set_option("pair_vrow", 1)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=every, delay=0, check=1)


@pytest.mark.parametrize("style,extra", [("table_ucgld", ()), ("table_ucg_bethe", ()), ("table_ucg_bethe", ("pseudo", "no")),
                                         ("table_ucg_bethe", ("method", "mf")),
                                         ("table_ucg_bethe", ("prior", "chemical_potential"))])
@pytest.mark.parametrize("ncell,tabstyle,tablength", [(12, "spline", 1024), (9, "linear", 1000), (7, "lookup", 900)])
def test_forces_bitwise_on_an_uploaded_list(fresh_ctx, pkg, orc, style, extra, ncell, tabstyle, tablength):
    """the oracle's beads, ghosts and full list on the device; 1728 / 729 / 343 beads = 4 / 2 / 1 blocks, the last ones ragged"""
    ctx = fresh_ctx
    ctx.set_option("pair_vrow", 1)
    deck = util.make_deck(tabstyle, tablength, extra_keywords=extra)
    beads = pkg.synth.make_beads(ncell, seed=ncell)
    if style == "table_ucg_bethe" and extra != ("prior", "chemical_potential"):
        beads.ucgp = np.clip(np.random.default_rng(3).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    op = util.oracle_pair(style, deck)
    op.set_sum_fixed(True)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    util.upload_from_oracle(ctx, sim, beads)
    gp = util.gpu_pair(ctx, style, deck)
    assert gp.sum_fixed
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    for k in ("f", "ucgforce", "scores"):
        assert util.bits_equal(G[k], O[k]), k
    ev = sim.ev()
    assert abs(eng - ev["eng_vdwl"]) <= 1e-12 * abs(ev["eng_vdwl"])
    assert np.allclose(vir, ev["virial"], rtol=1e-11, atol=1e-9)
    # and the reference's own order (half list, scatter, reverse sum) within the stated rounding
    sim0 = util.oracle_sim(beads, op, mode=0)
    sim0.rebuild()
    assert sim0.compute_forces(1, 1) == 0
    R = sim0.arrays()
    for k in ("f", "ucgforce", "scores"):
        if np.abs(R[k]).max() > 0:
            assert np.abs(G[k] - R[k]).max() <= REL_REFORDER * np.abs(R[k]).max(), k
    # the force-only launch gives the same bits as the energy / virial one
    ctx.force_clear()
    gp.compute(0, 0)
    G2 = ctx.atoms_download()
    for k in ("f", "ucgforce", "scores"):
        assert util.bits_equal(G2[k], G[k]), k


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
def test_the_option_applies_only_where_the_tables_fit(fresh_ctx, pkg, style):
    ctx = fresh_ctx
    ctx.set_option("pair_vrow", 1)
    for tabstyle, tablength, fixed in (("spline", 1024, True), ("spline", 2048, False), ("bitmap", 10, False)):
        gp = util.gpu_pair(ctx, style, util.make_deck(tabstyle, tablength))
        assert gp.sum_fixed == fixed, (tabstyle, tablength)
        gp.close()
    ctx.set_option("pair_vrow", 0)
    gp = util.gpu_pair(ctx, style, util.make_deck("spline", 1024))
    assert not gp.sum_fixed
    gp.close()


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
def test_several_actual_types(fresh_ctx, pkg, orc, style):
    deck = util.make_multi_deck(2, "spline", 256)
    beads = util.multi_type_beads(pkg, 9, 2, seed=17)
    beads.ucgp = np.clip(np.random.default_rng(3).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    op = util.oracle_pair_multi(style, deck)
    op.set_sum_fixed(True)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    ctx = fresh_ctx
    _setup_gpu(ctx, beads, 0.002, 1)
    ctx.neigh_rebuild()
    gp = util.gpu_pair_multi(ctx, style, deck)
    assert gp.sum_fixed
    gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["type"], O["type"])
    for k in ("f", "scores") + (("ucgforce",) if style == "table_ucgld" else ()):
        assert util.bits_equal(G[k], O[k]), k


@pytest.mark.parametrize("style,extra,langevin,ucgstate,integrator,steps,every",
                         [("table_ucgld", (), (1.0, 1.0, 1.0, 48279), "ld", "wall", 120, 1),
                          ("table_ucgld", (), (1.0, 1.0, 1.0, 48279), ("mc", 9127, 0.3), True, 100, 5),
                          ("table_ucg_bethe", (), None, ("mc", 4242, 0.3), True, 80, 2),
                          ("table_ucg_bethe", ("pseudo", "no"), None, "plain", True, 60, 1)])
def test_md_trajectory_bitwise(fresh_ctx, pkg, orc, style, extra, langevin, ucgstate, integrator, steps, every):
    """the resident loop (device rebuilds, hooks in the pair kernel's epilogue) on the virtual-row kernels: every array
    after `steps` steps with several re-neighbourings, bit for bit"""
    dt = 0.004
    deck = util.make_deck("spline", 1024, extra_keywords=extra)
    wall = integrator == "wall"
    beads = pkg.synth.make_beads(11, seed=31, ucgml=0.5 if wall else 10.0)  # 1331 beads: blocks of 512 / 512 / 307
    op = util.oracle_pair(style, deck)
    op.set_sum_fixed(True)
    sim = util.oracle_sim(beads, op, mode=1, dt=dt, langevin=langevin, nve=integrator, ucgstate=ucgstate, every=every)
    assert sim.setup(steps) == 0
    assert sim.run(steps, 10) == 0
    ctx = fresh_ctx
    _setup_gpu(ctx, beads, dt, every)
    gp = util.gpu_pair(ctx, style, deck)
    assert gp.sum_fixed
    if langevin is not None:
        ctx.fix_ucgld_langevin(*langevin)
    if ucgstate == "ld":
        ctx.fix_ucgstate("ld")
    elif ucgstate == "plain":
        ctx.fix_ucgstate(None)
    else:
        ctx.fix_ucgstate("mc", ucgstate[1], ucgstate[2])
    if wall:
        ctx.fix_nve_ucgld_wall_hard(False, 0.1)
    ctx.md_attach(gp, nve="wall" if wall else True, langevin=langevin is not None, ucgstate=True)
    ctx.md_setup(steps)
    ctx.md_run(steps, 10)
    gp.check_errors()
    info, oinfo = ctx.md_info(), sim.info()
    assert info["nrebuild"] == oinfo["nrebuild"] and info["nrebuild"] >= 3
    G, O = ctx.atoms_download(), sim.arrays()
    assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["ucgstate"], O["ucgstate"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(G[k], O[k]), k
    th, oth = ctx.md_thermo(), sim.ev()
    assert abs(th["eng_vdwl"] - oth["eng_vdwl"]) <= 1e-12 * abs(oth["eng_vdwl"])


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
@pytest.mark.parametrize("name", ["droplet", "gas", "two", "one"])
def test_ragged_and_empty_inputs(fresh_ctx, pkg, orc, style, name):
    """empty rows, a droplet in a mostly empty box, two beads, one bead: blocks with hardly any entries"""
    beads = _case(pkg, name)
    deck = util.make_deck("spline", 1024)
    op = util.oracle_pair(style, deck)
    op.set_sum_fixed(True)
    sim = util.oracle_sim(beads, op, mode=1, dt=0.002, nve=True, every=1)
    assert sim.setup(20) == 0
    ctx = fresh_ctx
    _setup_gpu(ctx, beads, 0.002, 1)
    gp = util.gpu_pair(ctx, style, deck)
    assert gp.sum_fixed
    ctx.md_attach(gp, nve=True, langevin=False, ucgstate=False)
    ctx.md_setup(20)
    G, O = ctx.atoms_download(), sim.arrays()
    for k in ("f", "scores", "ucgforce"):
        assert same(G[k], O[k]), k
    assert sim.run(20, 0) == 0
    ctx.md_run(20, 0)
    gp.check_errors()
    G, O = ctx.atoms_download(), sim.arrays()
    assert np.array_equal(G["tag"], O["tag"])
    for k in ("x", "v", "ucgl"):
        assert same(G[k], O[k]), k


def test_a_row_longer_than_the_builder_takes_is_reported(fresh_ctx, pkg):
    """a dense blob: rows of more than 128 entries.  The virtual-row builder refuses them with an error that names the
    way out (option pair_vrow 0); nothing is computed on partial lists"""
    ctx = fresh_ctx
    rng = np.random.default_rng(5)
    n = 400
    x = 6.0 + rng.uniform(-1.6, 1.6, (n, 3))  # 400 beads inside a 3.2^3 cube: everyone is everyone's neighbour
    from test_gpu_edges import _beads_from_points
    beads = _beads_from_points(pkg, x, 12.0)
    _setup_gpu(ctx, beads, 0.002, 1)
    ctx.neigh_rebuild()
    assert ctx.md_info()["maxrow"] > 128
    gp = util.gpu_pair(ctx, "table_ucgld", util.make_deck("spline", 1024))
    with pytest.raises(pkg.capi.UcgError) as ei:
        gp.compute(0, 0)
    assert "pair_vrow 0" in str(ei.value)
