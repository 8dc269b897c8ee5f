"""GPU parity against the oracle in LIBM mode.

The reference calls libm: std::exp / std::expm1 in the Bethe closure (UCG/pair_table_ucg_bethe.cpp:544-581), std::exp in
fix ucgstate (UCG/fix_ucgstate.cpp:88-132), tanh / log / exp in the density style.  The HIP kernels run the written
definition "ucg-math-v1" (csrc/ucg_math.h) instead, so that trajectories are reproducible bit for bit; every other GPU
test compares them with the oracle's OWN implementation of that definition (oracle/orc_math.c).  Here the oracle is
switched to glibc's functions -- what the reference itself executes on this box -- and the difference is bounded:

  * one force evaluation: forces, scores, posteriors within REL = 1e-12 of the largest magnitude of the field
    (a <= 1 ulp difference of exp / expm1 per pair, <= 4 ulp of tanh, through sums of ~50 terms);
  * whole trajectories (100 steps, re-neighbouring included): the discrete state of every bead at the end AND the
    number of beads whose state differs at any thermo checkpoint are reported and must be 0 -- a posterior would have
    to lie within an ulp of 1/2 (round) or of a 24-bit RanMars draw (mc) to flip; positions / lambda agree to 1e-9.
"""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

REL = 1e-12


class libm_oracle:
    """oracle math = libm inside the block, the written definition again afterwards (it is a global switch)"""

    def __init__(self, orc):
        self.L = orc.lib()

    def __enter__(self):
        self.L.orc_set_math(1)

    def __exit__(self, *a):
        self.L.orc_set_math(0)


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.mark.parametrize("style,extra", [("table_ucg_bethe", ()), ("table_ucg_bethe", ("pseudo", "no")),
                                         ("table_ucg_bethe", ("prior", "chemical_potential")),
                                         ("table_ucg_bethe_density", ())])
def test_one_evaluation_against_libm(gpu_ctx, pkg, orc, style, extra):
    dens = dict(density=(11.3, 1.5), extra11=0.05) if style.endswith("density") else {}
    deck = util.make_deck("spline", 1024, extra_keywords=extra, **dens)
    beads = pkg.synth.make_beads(10, seed=2025)
    rng = np.random.default_rng(9)
    beads.ucgp = np.clip(rng.uniform(size=beads.n), 1e-6, 1 - 1e-6)
    with libm_oracle(orc):
        op = util.oracle_pair(style, deck)  # prior_prob_from_type uses exp too
        sim = util.oracle_sim(beads, op, mode=1)
        sim.rebuild()
        A = util.upload_from_oracle(gpu_ctx, sim, beads)
        assert sim.compute_forces(1, 1) == 0
        O = sim.arrays()
        oev = sim.ev()
    gp = util.gpu_pair(gpu_ctx, style, deck)
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = gpu_ctx.atoms_download()
    worst = {k: _rel(G[k], O[k]) for k in ("f", "scores")}
    if style.endswith("density"):
        worst["ucgp"] = _rel(G["ucgp"], O["ucgp"])
    print("libm-mode differences:", worst)
    for k, v in worst.items():
        assert v <= REL, (k, v)
    assert abs(eng - oev["eng_vdwl"]) <= REL * abs(oev["eng_vdwl"])
    # it IS a different function: somewhere a bit differs (otherwise this test would not exercise anything)
    assert not (util.bits_equal(G["f"], O["f"]) and util.bits_equal(G["scores"], O["scores"]))
    assert A["nlocal"] == beads.n


@pytest.mark.parametrize("mode", ["plain", "ld", ("mc", 9127, 0.3)])
def test_fix_ucgstate_against_libm(gpu_ctx, pkg, orc, mode):
    """FixUCGState::post_force (UCG/fix_ucgstate.cpp:88-132) on the same scores: ucgp within 1e-15 (one exp each way and a
    division), identical states"""
    L = orc.lib()
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(10, seed=8)
    op = util.oracle_pair("table_ucgld", deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    gpu_ctx.set_units(1.0, 1.0, 1.0, 0.002)
    util.upload_from_oracle(gpu_ctx, sim, beads)
    gp = util.gpu_pair(gpu_ctx, "table_ucgld", deck)
    gp.compute(0, 0)
    assert sim.compute_forces(0, 0) == 0
    a = L.orc_sim_atoms(sim.h)
    if mode == "plain":
        st = L.orc_fix_ucgstate_create(0, 0, 0, 0.0, 0)
        gpu_ctx.fix_ucgstate(None)
    elif mode == "ld":
        st = L.orc_fix_ucgstate_create(1, 0, 0, 0.0, 0)
        gpu_ctx.fix_ucgstate("ld")
    else:
        st = L.orc_fix_ucgstate_create(0, 1, mode[1], mode[2], 0)
        gpu_ctx.fix_ucgstate("mc", mode[1], mode[2])
    with libm_oracle(orc):
        L.orc_fix_ucgstate_post_force(st, a)
    gpu_ctx.fix_ucgstate_post_force()
    G, O = gpu_ctx.atoms_download(), sim.arrays()
    assert np.abs(G["ucgp"] - O["ucgp"]).max() <= 1e-15
    assert np.array_equal(G["ucgstate"], O["ucgstate"])
    assert np.abs(G["ucgl"] - O["ucgl"]).max() <= 1e-15
    L.orc_fix_ucgstate_destroy(st)


TRAJ = [
    # style, extra keywords, ucgstate, dt, every
    ("table_ucg_bethe", ("method", "bethe", "pseudo", "yes", "prior", "ucgl"), "plain", 0.004, 2),
    ("table_ucg_bethe", ("pseudo", "no"), ("mc", 9127, 0.2), 0.004, 2),
    ("table_ucg_bethe_density", (), ("mc", 4242, 0.3), 0.002, 1),
    ("table_ucgld", (), ("mc", 777, 0.25), 0.004, 2),  # no math in the pair style: only fix ucgstate's exp differs
]


@pytest.mark.parametrize("style,extra,ucgstate,dt,every", TRAJ)
def test_state_trajectories_against_libm(fresh_ctx, pkg, orc, style, extra, ucgstate, dt, every):
    steps, chunk = 100, 10
    dens = dict(density=(11.3, 1.5), extra11=0.05) if style.endswith("density") else {}
    if ucgstate == "plain":
        dens["mu"] = (0.0, 0.0)  # posteriors around 1/2: round(ucgp) keeps both states populated and is at its most sensitive
    deck = util.make_deck("spline", 1024, extra_keywords=extra, **dens)
    beads = pkg.synth.make_beads(8, seed=31)
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, dt)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=every, delay=0, check=1)
    gp = util.gpu_pair(ctx, style, deck)
    if ucgstate == "plain":
        ctx.fix_ucgstate(None)
    else:
        ctx.fix_ucgstate("mc", ucgstate[1], ucgstate[2])
    ctx.md_attach(gp, nve=True, langevin=False, ucgstate=True)
    flips, total = 0, 0
    with libm_oracle(orc):
        op = util.oracle_pair(style, deck)
        sim = util.oracle_sim(beads, op, mode=1, dt=dt, langevin=None, nve=True, ucgstate=ucgstate, every=every)
        assert sim.setup(steps) == 0
        ctx.md_setup(steps)
        for _ in range(steps // chunk):
            assert sim.run(chunk, 0) == 0
            ctx.md_run(chunk, 0)
            G, O = ctx.atoms_download(), sim.arrays()
            assert np.array_equal(G["tag"], O["tag"])  # same re-neighbouring decisions, same order
            flips += int((G["ucgstate"] != O["ucgstate"]).sum())
            total += beads.n
    gp.check_errors()
    info, oinfo = ctx.md_info(), sim.info()
    print(f"{style} {ucgstate}: {flips} differing states in {total} bead-checkpoints over {steps} steps; "
          f"max |dx| = {np.abs(G['x'] - O['x']).max():.3e}, max |ducgp| = {np.abs(G['ucgp'] - O['ucgp']).max():.3e}")
    assert flips == 0
    assert info["nrebuild"] == oinfo["nrebuild"] >= 2
    for k in ("x", "v", "ucgl", "ucgp"):
        assert np.abs(G[k] - O[k]).max() <= 1e-9, k
    assert 0 < G["ucgstate"].sum() < beads.n
