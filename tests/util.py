"""helpers shared by the tests: build the same setup on the oracle and on the GPU library"""
import tempfile

import numpy as np

from conftest import load_oracle, load_package


def _tmpdir(prefix):
    """a scratch directory for generated decks, removed when the test process ends"""
    import atexit
    import shutil
    d = tempfile.mkdtemp(prefix=prefix)
    atexit.register(shutil.rmtree, d, True)
    return d


def make_deck(tabstyle="spline", tablength=1024, **kw):
    pkg = load_package()
    return pkg.synth.make_deck(_tmpdir("ucgdeck_"), tabstyle, tablength, **kw)


GATHER_SLOTS = 1  # the library's default lanes per bead (ucg_pair_gather_slots); part of the canonical order

# How a library pair sums a bead's terms -- ordered double sums (the default), or order-free fixed sums where option
# pair_vrow puts it on virtual rows (include/ucg_hip.h, ucg_pair_sum_fixed) -- is decided by the library at ucg_pair_init.
# The oracle states both; oracle pairs made here start with ordered sums, and a test that switches the option on puts its
# oracle pair into the same mode (op.set_sum_fixed(True)) and asserts gp.sum_fixed.


def oracle_pair(style, deck, T=1.0, ntypes=2, slots=GATHER_SLOTS):
    orc = load_oracle()
    p = orc.Pair(style)
    p.settings(deck.pair_style_args())
    p.coeff(deck.pair_coeff_args(), ntypes)
    p.init(ntypes, T, 1.0)
    p.set_gather_slots(slots)
    return p


def gpu_pair(ctx, style, deck, T=1.0, ntypes=2):
    pkg = load_package()
    p = pkg.capi.Pair(ctx, style)
    p.settings(deck.pair_style_args())
    p.coeff(deck.pair_coeff_args(), ntypes)
    p.init(ntypes, T)
    return p


def oracle_sim(beads, pair, mode=1, dt=0.002, langevin=None, nve=True, ucgstate=None, every=1, cutforce=2.5, skin=0.3):
    orc = load_oracle()
    s = orc.Sim(beads, cutforce, skin)
    s.set_run_params(dt=dt, every=every, delay=0, check=1, mode=mode)
    s.attach(pair, langevin=langevin, nve=nve, ucgstate=ucgstate)
    return s


def upload_from_oracle(ctx, sim, beads):
    """put the oracle's current owned+ghost beads and its full list on the GPU"""
    A = sim.arrays(ghosts=True)
    nl, ng = A["nlocal"], A["nghost"]
    ctx.atoms_upload(nl, ng, beads.ntypes, A["x"], A["v"], A["type"], A["tag"], A["mask"][:nl], A["ucgstate"],
                     A["ucgl"], A["ucgvl"], A["ucgml"], A["ucgp"], beads.mass)
    il, nn, fi, ne = sim.full_list()
    ctx.neigh_upload_full(nn, fi, ne)
    if ng:
        ctx.ghosts_upload(sim.ghost_map()[0])
    return A


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def max_ulp(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2**63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2**63) - ib[ib < 0]
    return int(np.abs(ia - ib).max()) if a.size else 0


def make_multi_deck(n_actual=2, tabstyle="spline", tablength=256, **kw):
    pkg = load_package()
    return pkg.synth.make_multi_deck(_tmpdir("ucgmdeck_"), n_actual, tabstyle, tablength, **kw)


def multi_type_beads(pkg, ncell, n_actual, seed, molecule_size=1):
    """beads of several actual atom types (random), masses for 2 * n_actual formal types, molecule ids"""
    beads = pkg.synth.make_beads(ncell, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    beads.ntypes = 2 * n_actual
    beads.mass = np.concatenate([[0.0], 1.0 + 0.25 * np.arange(2 * n_actual)])
    beads.molecule = ((beads.tag - 1) // molecule_size + 1).astype(np.int32)
    # one type per molecule, so that a molecule is wholly ON or OFF
    mtype = rng.integers(1, n_actual + 1, size=beads.molecule.max() + 1)
    beads.type = mtype[beads.molecule].astype(np.int32)
    return beads


def oracle_pair_multi(style, deck, T=1.0, slots=GATHER_SLOTS):
    orc = load_oracle()
    p = orc.Pair(style)
    p.settings(deck.pair_style_args())
    for cmd in deck.pair_coeff_commands():
        p.coeff(cmd, deck.ntypes)
    p.init(deck.ntypes, T, 1.0)
    p.set_gather_slots(slots)
    return p


def gpu_pair_multi(ctx, style, deck, T=1.0):
    pkg = load_package()
    p = pkg.capi.Pair(ctx, style)
    p.settings(deck.pair_style_args())
    for cmd in deck.pair_coeff_commands():
        p.coeff(cmd, deck.ntypes)
    p.init(deck.ntypes, T)
    return p
