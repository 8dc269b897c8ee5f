"""Resident path: device rebuild (wrap, sort, ghosts, bins, full list) and whole Verlet
trajectories on the GPU against the oracle, bit for bit (the "ucg-rebuild-v1" spec)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _setup_gpu(gpu_ctx, beads, dt, every):
    gpu_ctx.set_units(1.0, 1.0, 1.0, dt)
    gpu_ctx.upload_beads(beads)
    gpu_ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=every, delay=0, check=1)


@pytest.mark.parametrize("untiled", [0, 1])
@pytest.mark.parametrize("ncell", [5, 8, 14, 21])
def test_device_rebuild_matches_spec(gpu_ctx, pkg, orc, ncell, untiled):
    """both row builders (brick-tiled over LDS, and one lane per bead) against the specification"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(ncell, seed=ncell)
    # push a few beads out of the box so that the wrap is exercised
    beads.x[0] += beads.boxhi
    beads.x[1] -= beads.boxhi
    beads.x[2, 0] += beads.boxhi[0]
    op = util.oracle_pair("table_ucgld", deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    _setup_gpu(gpu_ctx, beads, 0.002, 1)
    gpu_ctx.set_option("rows_untiled", untiled)
    try:
        gpu_ctx.neigh_rebuild()
    finally:
        gpu_ctx.set_option("rows_untiled", 0)
    G = gpu_ctx.atoms_download(with_ghosts=True)
    O = sim.arrays(ghosts=True)
    assert (G["nlocal"], G["nghost"]) == (O["nlocal"], O["nghost"])
    nl = G["nlocal"]
    assert np.array_equal(G["tag"], O["tag"])
    assert util.bits_equal(G["x"], O["x"])
    assert np.array_equal(G["type"], O["type"]) and np.array_equal(G["ucgstate"], O["ucgstate"])
    assert util.bits_equal(G["ucgl"], O["ucgl"]) and util.bits_equal(G["ucgp"], O["ucgp"])
    assert util.bits_equal(G["v"], O["v"]) and util.bits_equal(G["ucgml"], O["ucgml"])
    gs, gsh = gpu_ctx.ghosts_download()
    os_, osh = sim.ghost_map()
    assert np.array_equal(gs, os_) and np.array_equal(gsh, osh)
    gl = gpu_ctx.neigh_download()
    ol = sim.full_list()
    for a, b in zip(gl, ol):
        assert np.array_equal(a, b)
    # and the forces on that list
    gp = util.gpu_pair(gpu_ctx, "table_ucgld", deck)
    gp.compute(0, 0)
    assert sim.compute_forces(0, 0) == 0
    assert util.bits_equal(gpu_ctx.atoms_download()["f"], sim.arrays()["f"])
    assert nl == beads.n


def test_tiled_row_builder_with_rows_longer_than_its_byte_counters(gpu_ctx, pkg, orc):
    """a dense clump inside a dilute melt: rows of more than 255 entries (the tiled builder counts a row's distance classes in
    byte fields while it walks the candidates and recounts such a row from its column), candidates still within the staging
    capacity of a brick: the rows are the specification's, entry by entry, and the tiled builder did build them"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(12, seed=4)
    rng = np.random.default_rng(99)
    centre = 0.5 * (beads.boxlo + beads.boxhi)
    m = 330
    d = rng.normal(size=(m, 3))
    d *= (1.2 * rng.random(m) ** (1.0 / 3.0) / np.linalg.norm(d, axis=1))[:, None]
    beads.x[:m] = centre + d  # 330 beads inside a ball of radius 1.2: all within the list cutoff of one another
    op = util.oracle_pair("table_ucgld", deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    _setup_gpu(gpu_ctx, beads, 0.002, 1)
    gpu_ctx.neigh_rebuild()
    info = gpu_ctx.md_info()
    gl = gpu_ctx.neigh_download()
    ol = sim.full_list()
    assert int(ol[1].max()) >= 300 and info["maxrow"] == int(ol[1].max())
    for a, b in zip(gl, ol):
        assert np.array_equal(a, b)
    # the same rows from the one-lane-per-bead builder
    gpu_ctx.set_option("rows_untiled", 1)
    try:
        gpu_ctx.upload_beads(beads)
        gpu_ctx.neigh_rebuild()
    finally:
        gpu_ctx.set_option("rows_untiled", 0)
    for a, b in zip(gpu_ctx.neigh_download(), ol):
        assert np.array_equal(a, b)


CASES = [
    # style, extra keywords, langevin, ucgstate, dt, steps, every, integrator
    ("table_ucgld", (), (1.0, 1.0, 1.0, 48279), "ld", 0.004, 120, 1, True),
    ("table_ucgld", (), (1.0, 2.0, 0.5, 777), "ld", 0.004, 100, 5, True),
    ("table_ucg_bethe", (), None, "plain", 0.004, 80, 1, True),
    ("table_ucg_bethe", ("pseudo", "no"), None, ("mc", 9127, 0.2), 0.004, 80, 2, True),
    ("table_ucgld", (), None, None, 0.004, 60, 1, True),
    ("table_ucg_bethe_density", (), None, ("mc", 4242, 0.3), 0.002, 60, 1, True),
    ("table_ucg_bethe_density", (), None, None, 0.002, 40, 2, True),
    # fix nve/ucgld/wall/hard: states follow lambda, lambda is reflected at 0 and 1 (a light
    # lambda mass and a hot thermostat make every bead hit the walls), with and without the bias
    ("table_ucgld", (), (6.0, 6.0, 0.2, 48279), "ld", 0.004, 120, 1, "wall"),
    ("table_ucgld", (), (6.0, 6.0, 0.2, 1234), "ld", 0.004, 100, 2, ("wall", 0.25)),
    ("table_ucgld", (), None, None, 0.004, 60, 1, ("wall", 0.1)),
    # tabstyle bitmap (the "tab" keyword below is consumed by the test, not by the pair style)
    ("table_ucgld", ("tab", "bitmap", 12), (1.0, 1.0, 1.0, 48279), "ld", 0.004, 80, 1, True),
    ("table_ucg_bethe", ("tab", "bitmap", 11), None, ("mc", 9127, 0.2), 0.004, 60, 2, True),
    ("table_ucg_bethe_density", ("tab", "bitmap", 12), None, ("mc", 4242, 0.3), 0.002, 40, 1, True),
]


@pytest.mark.parametrize("style,extra,langevin,ucgstate,dt,steps,every,integrator", CASES)
def test_md_trajectory_bitwise(fresh_ctx, pkg, orc, style, extra, langevin, ucgstate, dt, steps, every, integrator):
    dens = dict(density=(11.3, 1.5), extra11=0.05) if style.endswith("density") else {}
    tabstyle, tablength = "spline", 1024
    if extra[:1] == ("tab",):
        tabstyle, tablength, extra = extra[1], extra[2], extra[3:]
    deck = util.make_deck(tabstyle, tablength, extra_keywords=extra, **dens)
    wall = integrator is not True
    beads = pkg.synth.make_beads(8, seed=31, ucgml=0.5 if wall else 10.0)
    op = util.oracle_pair(style, deck)
    sim = util.oracle_sim(beads, op, mode=1, dt=dt, langevin=langevin, nve=integrator, ucgstate=ucgstate, every=every)
    assert sim.setup(steps) == 0
    assert sim.run(steps, 10) == 0

    gpu_ctx = fresh_ctx
    _setup_gpu(gpu_ctx, beads, dt, every)
    gp = util.gpu_pair(gpu_ctx, style, deck)
    if langevin is not None:
        gpu_ctx.fix_ucgld_langevin(*langevin)
    if ucgstate is not None:
        if ucgstate == "ld":
            gpu_ctx.fix_ucgstate("ld")
        elif ucgstate == "plain":
            gpu_ctx.fix_ucgstate(None)
        else:
            gpu_ctx.fix_ucgstate("mc", ucgstate[1], ucgstate[2])
    if wall:
        bias = isinstance(integrator, tuple)
        gpu_ctx.fix_nve_ucgld_wall_hard(bias, integrator[1] if bias else 0.1)
    gpu_ctx.md_attach(gp, nve="wall" if wall else True, langevin=langevin is not None, ucgstate=ucgstate is not None)
    gpu_ctx.md_setup(steps)
    gpu_ctx.md_run(steps, 10)
    gp.check_errors()
    info, oinfo = gpu_ctx.md_info(), sim.info()
    assert info["nrebuild"] == oinfo["nrebuild"] and info["nrebuild"] >= 2
    assert info["nghost"] == oinfo["nghost"] and info["list_entries"] == oinfo["nfull"]
    G = gpu_ctx.atoms_download()
    O = sim.arrays()
    assert np.array_equal(G["tag"], O["tag"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(G[k], O[k]), k
    assert np.array_equal(G["ucgstate"], O["ucgstate"])
    th, oth = gpu_ctx.md_thermo(), sim.ev()
    assert abs(th["eng_vdwl"] - oth["eng_vdwl"]) <= 1e-12 * abs(oth["eng_vdwl"])
    if langevin is not None:
        assert abs(th["lambda_temp"] - oth["lambda_temp"]) <= 1e-12 * abs(oth["lambda_temp"])
    if isinstance(ucgstate, tuple):  # mc: both states stay populated
        assert 0 < G["ucgstate"].sum() < beads.n
    if wall:
        # lambda stays between the walls (up to one step's overshoot) and both states are populated
        assert np.all(G["ucgl"] > -0.3) and np.all(G["ucgl"] < 1.3)
        assert 0 < G["ucgstate"].sum() < beads.n


@pytest.mark.parametrize("options", [dict(post_in_pair=0), dict(post_in_pair=0, md_no_fuse=1), dict(gather_slots=4),
                                     dict(pair_vrow=1), dict(pair_vrow=1, post_in_pair=0)])
@pytest.mark.parametrize("style,ucgstate", [("table_ucgld", "ld"), ("table_ucg_bethe", ("mc", 9127, 0.2))])
def test_resident_loop_variants_give_the_same_bits(fresh_ctx, pkg, orc, style, ucgstate, options):
    """the resident loop has three forms of an ordinary step -- hooks in the gather kernel's epilogue (default), pair
    kernel + fused per-bead kernel (post_in_pair = 0), and that with initial_integrate as its own launch
    (md_no_fuse = 1): all must reproduce the oracle bit for bit (the other trajectory tests run the default) -- on the
    full-row gather kernels (ordered sums, whose order the lanes per bead are part of) and on the virtual-row kernels
    (option pair_vrow: fixed sums)"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(8, seed=77)
    lang = (1.0, 1.0, 1.0, 48279) if style == "table_ucgld" else None
    slots = options.get("gather_slots", 1)
    op = util.oracle_pair(style, deck, slots=slots)
    op.set_sum_fixed(options.get("pair_vrow", 0) == 1)
    sim = util.oracle_sim(beads, op, mode=1, dt=0.004, langevin=lang, nve=True, ucgstate=ucgstate, every=2)
    assert sim.setup(50) == 0 and sim.run(50, 20) == 0
    ctx = fresh_ctx
    _setup_gpu(ctx, beads, 0.004, 2)
    for k, v in options.items():
        ctx.set_option(k, v)
    gp = util.gpu_pair(ctx, style, deck)
    assert gp.sum_fixed == (options.get("pair_vrow", 0) == 1)
    if lang:
        ctx.fix_ucgld_langevin(*lang)
    if ucgstate == "ld":
        ctx.fix_ucgstate("ld")
    else:
        ctx.fix_ucgstate("mc", ucgstate[1], ucgstate[2])
    ctx.md_attach(gp, nve=True, langevin=lang is not None, ucgstate=True)
    ctx.md_setup(50)
    ctx.md_run(50, 20)
    gp.check_errors()
    G, O = ctx.atoms_download(), sim.arrays()
    assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["ucgstate"], O["ucgstate"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(G[k], O[k]), k


@pytest.mark.parametrize("hot_block", [1, 0])
@pytest.mark.parametrize("style,tabstyle,tablength", [("table_ucgld", "spline", 2048), ("table_ucg_bethe", "spline", 4096),
                                                      ("table_ucgld", "linear", 6000)])
def test_tables_longer_than_the_lds_trajectories_bitwise(fresh_ctx, pkg, orc, style, tabstyle, tablength, hot_block):
    """one actual type whose tables do not fit the LDS: the kernels read them through L2 and -- option hot_block, the
    default -- keep the far end of the r^2 grid in LDS (PairDev::hot_k0); trajectories equal the oracle's bit for bit
    with the window and without it"""
    extra = ("method", "bethe", "pseudo", "yes", "prior", "ucgl") if style == "table_ucg_bethe" else ()
    deck = util.make_deck(tabstyle, tablength, extra_keywords=extra)
    beads = pkg.synth.make_beads(8, seed=31)
    lang = (1.0, 1.0, 1.0, 48279) if style == "table_ucgld" else None
    op = util.oracle_pair(style, deck)
    sim = util.oracle_sim(beads, op, mode=1, dt=0.004, langevin=lang, nve=True, ucgstate="ld" if lang else "plain", every=2)
    assert sim.setup(40) == 0 and sim.run(40, 20) == 0
    ctx = fresh_ctx
    ctx.set_option("hot_block", hot_block)
    _setup_gpu(ctx, beads, 0.004, 2)
    gp = util.gpu_pair(ctx, style, deck)
    if lang:
        ctx.fix_ucgld_langevin(*lang)
        ctx.fix_ucgstate("ld")
    else:
        ctx.fix_ucgstate(None)
    ctx.md_attach(gp, nve=True, langevin=lang is not None, ucgstate=True)
    ctx.md_setup(40)
    ctx.md_run(40, 20)
    gp.check_errors()
    G, O = ctx.atoms_download(), sim.arrays()
    assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["ucgstate"], O["ucgstate"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(G[k], O[k]), k


@pytest.mark.parametrize("style,ncell,steps,vrow", [("table_ucgld", 64, 16, 0), ("table_ucg_bethe", 64, 16, 0), ("table_ucgld", 100, 12, 0),
                                                    ("table_ucg_bethe_density", 64, 12, 0), ("table_ucgld", 64, 16, 1),
                                                    ("table_ucg_bethe", 64, 16, 1)])
def test_baseline_size_trajectory_bitwise_vs_oracle(fresh_ctx, pkg, orc, style, ncell, steps, vrow):
    """BASELINE.json config 2 at its full size -- 262 144 beads (64^3), pair_table_ucgld + fix ucgld/langevin (+ fix
    ucgstate ld, fix nve/ucgld/wall/hard) -- against the ORACLE, bit for bit, over a re-neighbouring: 256 workgroups of the
    gather kernel (one full round of the chip), ~1800 bricks of the row builder, 69 k periodic images.  The Bethe style
    runs the same size once, and the headline workload (1 000 000 beads: 977 workgroups, 164 k images) a dozen steps.
    vrow = 1: the same on the virtual-row kernels (option pair_vrow: 512 workgroups, five lists each) against the
    oracle's fixed sums."""
    bethe, dens = style == "table_ucg_bethe", style == "table_ucg_bethe_density"
    kw = dict(density=(11.3, 1.5), extra11=0.05) if dens else {}
    deck = util.make_deck("spline", 1024, extra_keywords=("method", "bethe", "pseudo", "yes", "prior", "ucgl") if bethe else (), **kw)
    beads = pkg.synth.make_beads(ncell, seed=12345)
    assert beads.n == ncell ** 3
    lang = None if (bethe or dens) else (1.0, 1.0, 1.0, 48279)
    ust = ("mc", 9127, 0.01) if dens else ("plain" if bethe else "ld")
    # at dt = 0.004 the fastest of these beads crosses half the skin after ~8 steps
    op = util.oracle_pair(style, deck)
    op.set_sum_fixed(bool(vrow))
    sim = util.oracle_sim(beads, op, mode=1, dt=0.004, langevin=lang, nve="wall", ucgstate=ust, every=2)
    assert sim.setup(steps) == 0 and sim.run(steps, 0) == 0
    ctx = fresh_ctx
    _setup_gpu(ctx, beads, 0.004, 2)
    ctx.set_option("pair_vrow", vrow)
    gp = util.gpu_pair(ctx, style, deck)
    assert gp.sum_fixed == bool(vrow)
    if lang:
        ctx.fix_ucgld_langevin(*lang)
        ctx.fix_ucgstate("ld")
    elif dens:
        ctx.fix_ucgstate("mc", 9127, 0.01)
    else:
        ctx.fix_ucgstate(None)
    ctx.fix_nve_ucgld_wall_hard(False, 0.1)
    ctx.md_attach(gp, nve="wall", langevin=lang is not None, ucgstate=True)
    ctx.md_setup(steps)
    ctx.md_run(steps, 0)
    gp.check_errors()
    info, oinfo = ctx.md_info(), sim.info()
    assert info["nrebuild"] == oinfo["nrebuild"] >= 2  # setup + at least one re-neighbouring
    assert info["nghost"] == oinfo["nghost"] and info["list_entries"] == oinfo["nfull"]
    G, O = ctx.atoms_download(), sim.arrays()
    assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["ucgstate"], O["ucgstate"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(G[k], O[k]), k


def test_md_run_reports_table_range_errors(fresh_ctx, pkg):
    """a bead driven inside the tables' inner cutoff DURING a resident run: ucg_md_run itself returns the reference's
    error (UCG/pair_table_ucgld.cpp:436-444, error->one) -- the sticky device flag is polled at every re-neighbouring,
    on thermo steps and after the last step, not only when the caller remembers ucg_pair_check_errors"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(8, seed=3)
    # bead 0 and its nearest neighbour approach head-on, far too fast for the repulsion to stop them: ~1 apart, 0.48
    # closer after every step of 0.004 at relative speed 120, so inside the tables' inner cutoff 0.6 within two steps
    d = beads.x - beads.x[0]
    d -= np.round(d / beads.boxhi) * beads.boxhi
    r = np.sqrt((d * d).sum(axis=1))
    r[0] = 1e9
    j = int(np.argmin(r))
    u = d[j] / r[j]
    beads.v[0] = 60.0 * u
    beads.v[j] = -60.0 * u
    beads.ucgml[:] = 1.0e9  # lambda stays put
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.004)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    gp = util.gpu_pair(ctx, "table_ucgld", deck)
    ctx.md_attach(gp, nve=True, langevin=False, ucgstate=False)
    ctx.md_setup(200)
    with pytest.raises(pkg.capi.UcgError) as ei:
        ctx.md_run(200, 0)
    assert ei.value.code in (4, 5) and "table" in str(ei.value)
    assert ctx.md_info()["ntimestep"] < 200  # stopped at a re-neighbouring, not at the end
