"""Several actual atom types (the kernels' general path) and fix cluster_switch on the GPU against the
oracle: forces bit for bit, cluster labels / molecule states / accept flags / atom types equal, and a
whole trajectory with switching every few steps bit for bit."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _setup_gpu(ctx, beads, dt, every):
    ctx.set_units(1.0, 1.0, 1.0, dt)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=every, delay=0, check=1)


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
@pytest.mark.parametrize("n_actual,tablength,hot_block,kind_blocks",
                         [(2, 256, 1, 1), (3, 128, 1, 1), (2, 1024, 1, 1), (2, 1024, 0, 1), (3, 1024, 1, 1), (2, 1024, 1, 0), (3, 1024, 0, 0)])
def test_several_actual_types_forces_bitwise(fresh_ctx, pkg, orc, style, n_actual, tablength, hot_block, kind_blocks):
    """tables per actual pair, cutoffs and mu per type: the general (not one-type) path of the pair kernels;
    2 x 1024-knot decks do not fit LDS and read their tables through L2, with (option hot_block, the default) the block
    of the most populous type in LDS next to that path and (option kind_blocks, the default) the cold lanes reading a compact
    block per (row type, neighbour type) kind instead of the full layout"""
    deck = util.make_multi_deck(n_actual, "spline", tablength)
    beads = util.multi_type_beads(pkg, 9, n_actual, seed=17)
    beads.ucgp = np.clip(np.random.default_rng(3).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    op = util.oracle_pair_multi(style, deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    assert len(np.unique(O["type"])) == n_actual
    ctx = fresh_ctx
    ctx.set_option("hot_block", hot_block)
    ctx.set_option("kind_blocks", kind_blocks)
    _setup_gpu(ctx, beads, 0.002, 1)
    ctx.neigh_rebuild()
    gp = util.gpu_pair_multi(ctx, style, deck)
    out = gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["type"], O["type"])
    for k in ("f", "scores") + (("ucgforce",) if style == "table_ucgld" else ()):
        assert util.bits_equal(G[k], O[k]), k
    ev = sim.ev()
    assert abs(out[0] - ev["eng_vdwl"]) <= 1e-12 * abs(ev["eng_vdwl"])
    # reference order (half list + scatter) agrees to rounding
    sim0 = util.oracle_sim(beads, op, mode=0)
    sim0.rebuild()
    assert sim0.compute_forces(0, 0) == 0
    assert np.max(np.abs(sim0.arrays()["f"] - G["f"])) < 1e-10


@pytest.mark.parametrize("hot_block,kind_blocks", [(1, 1), (0, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("tablength", [256, 1024])
def test_several_actual_types_density_style_bitwise(fresh_ctx, pkg, orc, tablength, hot_block, kind_blocks):
    """table_ucg_bethe_density on two actual types (BASELINE config 5's deck): 256 knots fit the LDS; with 1024 knots
    the ten tables are read through L2 -- from a compact block per (row type, neighbour type) kind (option kind_blocks,
    the default) or from the full layout -- and (option hot_block, the default) the block of the most populous type is
    staged in LDS next to that path.  Bit for bit the oracle's canonical order every way."""
    deck = util.make_multi_deck(2, "spline", tablength, density=(11.3, 1.5), extra11=0.05, n_file=2000)
    beads = util.multi_type_beads(pkg, 9, 2, seed=23)
    op = util.oracle_pair_multi("table_ucg_bethe_density", deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    ctx = fresh_ctx
    ctx.set_option("hot_block", hot_block)
    ctx.set_option("kind_blocks", kind_blocks)
    util.upload_from_oracle(ctx, sim, beads)
    gp = util.gpu_pair_multi(ctx, "table_ucg_bethe_density", deck)
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    assert len(np.unique(O["type"])) == 2 and not np.isnan(G["f"]).any()
    for k in ("f", "scores", "ucgp"):
        assert util.bits_equal(G[k], O[k]), k
    ev = sim.ev()
    assert abs(eng - ev["eng_vdwl"]) <= 1e-12 * abs(ev["eng_vdwl"])
    # and without energies (the kernels of an ordinary step)
    gp.compute(0, 0)
    G2 = ctx.atoms_download()
    assert util.bits_equal(G2["f"], O["f"]) and util.bits_equal(G2["scores"], O["scores"])


def _cluster_case(pkg, ncell, seed, molecule_size, prob_on, cutoff):
    deck = util.make_multi_deck(2, "spline", 256)
    beads = util.multi_type_beads(pkg, ncell, 2, seed=seed, molecule_size=molecule_size)
    rates, contacts = pkg.synth.write_cluster_switch_files(deck.workdir, prob_on, [1], [2], [(1, 1)])
    mol_seed = int(beads.molecule[np.flatnonzero(beads.type == 1)[0]])  # an ON molecule
    return deck, beads, rates, contacts, mol_seed


@pytest.mark.parametrize("molecule_size,cutoff", [(1, 1.25), (2, 1.15), (4, 1.05)])
def test_cluster_switch_check_and_attempt(fresh_ctx, pkg, orc, molecule_size, cutoff):
    deck, beads, rates, contacts, mol_seed = _cluster_case(pkg, 10, 5, molecule_size, 0.35, cutoff)
    op = util.oracle_pair_multi("table_ucgld", deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    sim.cluster_switch(mol_seed, 0, cutoff, 4711, 5, rates, contacts)
    ctx = fresh_ctx
    _setup_gpu(ctx, beads, 0.002, 1)
    ctx.neigh_rebuild()
    ctx.fix_cluster_switch(mol_seed, 0, cutoff, 4711, 5, rates, contacts)
    assert np.array_equal(ctx.download_molecule(), sim.arrays()["molecule"])
    L = orc.lib()
    for rep in range(3):
        cs = L.orc_sim_cs(sim.h)
        assert L.orc_cs_check_cluster(cs, L.orc_sim_atoms(sim.h), L.orc_sim_molecule(sim.h), L.orc_sim_full_list(sim.h)) == 0
        ctx.fix_cluster_switch_check_cluster()
        A, B = ctx.fix_cluster_switch_arrays(), sim.cs_arrays()
        for k in ("mol_cluster", "mol_state", "mol_restrict"):
            assert np.array_equal(A[k], B[k]), (rep, k)
        ncl = int((B["mol_cluster"] == B["mol_cluster"][mol_seed]).sum())
        if rep == 0:
            assert 1 < ncl < beads.molecule.max()  # a real cluster, not everything
        assert L.orc_cs_attempt_switch(cs, L.orc_sim_atoms(sim.h), L.orc_sim_molecule(sim.h)) == 0
        ctx.fix_cluster_switch_attempt_switch()
        A, B = ctx.fix_cluster_switch_arrays(), sim.cs_arrays()
        for k in ("mol_state", "mol_restrict", "mol_accept"):
            assert np.array_equal(A[k], B[k]), (rep, k)
        assert (B["mol_accept"] == 1).sum() > 0
        assert np.array_equal(ctx.atoms_download()["type"], sim.arrays()["type"])
        assert np.array_equal(ctx.fix_cluster_switch_vector(), sim.cs_stats())
        # a molecule is wholly ON or OFF after the flips
        t, m = sim.arrays()["type"], sim.arrays()["molecule"]
        for mol in np.unique(m)[:50]:
            assert len(np.unique(t[m == mol])) == 1


@pytest.mark.parametrize("style,freq", [("table_ucgld", 7), ("table_ucg_bethe", 4)])
def test_md_with_cluster_switch_bitwise(fresh_ctx, pkg, orc, style, freq):
    deck, beads, rates, contacts, mol_seed = _cluster_case(pkg, 8, 9, 2, 0.4, 1.2)
    steps, dt = 60, 0.004
    op = util.oracle_pair_multi(style, deck)
    lang = (1.0, 1.0, 1.0, 48279) if style == "table_ucgld" else None
    ucgst = "ld" if style == "table_ucgld" else "plain"
    sim = util.oracle_sim(beads, op, mode=1, dt=dt, langevin=lang, nve=True, ucgstate=ucgst, every=5)
    sim.cluster_switch(mol_seed, 0, 1.2, 99, freq, rates, contacts)
    assert sim.setup(steps) == 0
    assert sim.run(steps, 10) == 0
    ctx = fresh_ctx
    _setup_gpu(ctx, beads, dt, 5)
    gp = util.gpu_pair_multi(ctx, style, deck)
    if lang:
        ctx.fix_ucgld_langevin(*lang)
    ctx.fix_ucgstate("ld" if ucgst == "ld" else None)
    ctx.fix_cluster_switch(mol_seed, 0, 1.2, 99, freq, rates, contacts)
    ctx.md_attach(gp, nve=True, langevin=lang is not None, ucgstate=True)
    ctx.md_setup(steps)
    ctx.md_run(steps, 10)
    gp.check_errors()
    assert ctx.md_info()["nrebuild"] == sim.info()["nrebuild"] >= steps // freq
    G, O = ctx.atoms_download(), sim.arrays()
    assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["type"], O["type"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "scores", "ucgp"):
        assert util.bits_equal(G[k], O[k]), k
    assert np.array_equal(ctx.fix_cluster_switch_vector(), sim.cs_stats())
    assert sim.cs_stats()[1] > 0  # some switches were accepted
