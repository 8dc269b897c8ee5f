"""Edge cases and BASELINE-size properties of the resident path.

Edge cases (against the oracle, bit for bit): empty rows (isolated beads), a droplet in a box much
larger than itself (most bins and bricks empty), two beads, one bead.
Full size (1,000,000 beads, no oracle: size-independent properties): Newton's third law (the forces sum
to rounding), every pair listed from both ends, sorted order, idempotent rebuild, run-to-run bitwise
reproducibility."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def same(a, b):
    """equal bit patterns; where the reference arithmetic gives NaN (the density style's unguarded closure on
    beads without neighbours) both must be NaN -- NaN payloads are hardware business"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    ok = ~np.isnan(a)
    return np.array_equal(a[ok].view(np.uint64), b[ok].view(np.uint64))


def _beads_from_points(pkg, x, box, seed=3):
    n = len(x)
    rng = np.random.default_rng(seed)
    b = pkg.synth.make_beads(2, seed=seed)
    return pkg.synth.Beads(
        n=n, boxlo=np.zeros(3), boxhi=np.full(3, float(box)), x=np.ascontiguousarray(x, dtype=np.float64),
        v=rng.normal(0.0, 1.0, (n, 3)), type=np.ones(n, dtype=np.int32), tag=np.arange(1, n + 1, dtype=np.int32),
        mask=np.ones(n, dtype=np.int32), ucgstate=(rng.uniform(size=n) < 0.5).astype(np.int32), ucgl=rng.uniform(size=n),
        ucgvl=np.zeros(n), ucgml=np.full(n, 10.0), ucgp=np.full(n, -1.0), mass=b.mass)


def _case(pkg, name):
    if name == "droplet":            # 300 beads in a 40^3 box: almost every bin, and most bricks, are empty
        return pkg.synth.make_cluster(300, box=40.0, radius=4.0, seed=2)
    if name == "gas":                # isolated beads (empty rows) plus a few close pairs
        rng = np.random.default_rng(8)
        g = np.stack(np.meshgrid(*[np.arange(6)] * 3, indexing="ij"), -1).reshape(-1, 3) * 5.0 + 2.5
        x = g + rng.uniform(-0.5, 0.5, g.shape)
        extra = x[:20] + np.array([1.05, 0.0, 0.0])
        return _beads_from_points(pkg, np.concatenate([x, extra]), 30.0)
    if name == "two":
        return _beads_from_points(pkg, np.array([[5.0, 5.0, 5.0], [6.1, 5.2, 4.9]]), 12.0)
    if name == "one":
        return _beads_from_points(pkg, np.array([[3.0, 4.0, 5.0]]), 12.0)
    raise ValueError(name)


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe", "table_ucg_bethe_density"])
@pytest.mark.parametrize("name", ["droplet", "gas", "two", "one"])
def test_ragged_and_empty_inputs(fresh_ctx, pkg, orc, style, name):
    beads = _case(pkg, name)
    dens = dict(density=(11.3, 1.5), extra11=0.05) if style.endswith("density") else {}
    deck = util.make_deck("spline", 1024, **dens)
    op = util.oracle_pair(style, deck)
    sim = util.oracle_sim(beads, op, mode=1, dt=0.002, nve=True, every=1)
    assert sim.setup(20) == 0
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    gp = util.gpu_pair(ctx, style, deck)
    ctx.md_attach(gp, nve=True, langevin=False, ucgstate=False)
    ctx.md_setup(20)
    gl, ol = ctx.neigh_download(), sim.full_list()
    for a, b in zip(gl, ol):
        assert np.array_equal(a, b)
    if name in ("gas", "one"):
        assert (gl[1] == 0).sum() > 0  # empty rows exist
    G, O = ctx.atoms_download(), sim.arrays()
    for k in ("f", "scores", "ucgforce"):
        assert same(G[k], O[k]), k
    assert sim.run(20, 0) == 0
    ctx.md_run(20, 0)
    gp.check_errors()
    G, O = ctx.atoms_download(), sim.arrays()
    assert np.array_equal(G["tag"], O["tag"])
    for k in ("x", "v", "f", "ucgl", "ucgvl"):
        assert same(G[k], O[k]), k


def test_full_size_properties_1M_beads(fresh_ctx, pkg):
    """BASELINE.json's size: 100^3 beads at rho* = 0.8, the bench workload"""
    beads = pkg.synth.make_beads(100, seed=12345)
    deck = util.make_deck("spline", 1024)
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.002)

    def run(nsteps):
        nre0 = ctx.md_info()["nrebuild"] if ran else 0
        ran.append(1)
        ctx.upload_beads(beads)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
        gp = util.gpu_pair(ctx, "table_ucgld", deck)
        ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279)
        ctx.fix_ucgstate("ld")
        ctx.fix_nve_ucgld_wall_hard(False, 0.1)
        ctx.md_attach(gp, nve="wall", langevin=True, ucgstate=True)
        ctx.md_setup(nsteps)
        first = ctx.atoms_download()
        lst = ctx.neigh_download() + (ctx.md_info()["list_entries"],)
        ctx.md_run(nsteps, 0)
        gp.check_errors()
        info = ctx.md_info()
        info["nrebuild"] -= nre0
        return first, ctx.atoms_download(), lst, info

    ran = []
    A0, A1, (il, nn, fi, ne, entries0), info = run(25)
    n = beads.n
    assert A0["nlocal"] == n and sorted(A0["tag"][:1000].tolist()) != A0["tag"][:1000].tolist()  # Morton order, not tag order
    assert np.array_equal(np.sort(A0["tag"]), np.arange(1, n + 1))
    # Newton's third law on the gathered forces: sums vanish to rounding
    f = A0["f"]
    assert np.max(np.abs(f.sum(axis=0))) < 1e-9 * np.abs(f).sum() / n * np.sqrt(n) + 1e-7
    # full list: every owned-owned pair appears from both ends, total = 2 x half list
    idx = ne & 0x1FFFFFFF
    assert int(nn.sum()) == len(ne) == entries0 and len(ne) % 2 == 0
    rows = np.repeat(np.arange(n), nn)
    owned = idx < n
    a, b = rows[owned].astype(np.int64), idx[owned].astype(np.int64)
    assert np.array_equal(np.sort(a * n + b), np.sort(b * n + a))
    # the orientation bit is antisymmetric on owned pairs (exactly one end plays the reference's "i")
    orient = ((ne >> 29) & 1)[owned]
    key_ab, key_ba = a * n + b, b * n + a
    o1 = orient[np.argsort(key_ab)]
    o2 = orient[np.argsort(key_ba)]
    assert np.all(o1 + o2 == 1)
    # 72 neighbours per bead at rho* = 0.8, list radius 2.8
    assert 65.0 < len(ne) / n < 85.0  # (4/3) pi 2.8^3 x 0.8 = 73.6 for a uniform fluid; the jittered lattice has shells
    # reproducibility: the same run again gives the same bits; rebuilds happened in between
    B0, B1, _, info2 = run(25)
    assert info2["nrebuild"] == info["nrebuild"] >= 2
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgp", "scores"):
        assert util.bits_equal(A1[k], B1[k]), k
    assert np.array_equal(A1["tag"], B1["tag"])
    # the hard walls keep lambda in [0, 1]; posteriors stay inside the clamp of fix ucgstate
    assert np.all((A1["ucgl"] >= 0.0) & (A1["ucgl"] <= 1.0)) and np.all((A1["ucgp"] >= 1e-6) & (A1["ucgp"] <= 1 - 1e-6))


def test_full_size_properties_1M_beads_bethe(fresh_ctx, pkg):
    """BASELINE.json config 3 at its full size: 1,000,000 beads, table_ucg_bethe (method bethe, pseudo yes, prior ucgl)
    + fix ucgstate + fix nve/ucgld/wall/hard.  No oracle at this size: Newton's third law, posteriors inside the clamp of
    fix ucgstate, states = round(posterior), and run-to-run bitwise reproducibility across re-neighbourings."""
    beads = pkg.synth.make_beads(100, seed=12345)
    deck = util.make_deck("spline", 1024, extra_keywords=("method", "bethe", "pseudo", "yes", "prior", "ucgl"))
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ran = []

    def run(nsteps):
        nre0 = ctx.md_info()["nrebuild"] if ran else 0
        ran.append(1)
        ctx.upload_beads(beads)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
        gp = util.gpu_pair(ctx, "table_ucg_bethe", deck)
        ctx.fix_ucgstate(None)
        ctx.fix_nve_ucgld_wall_hard(False, 0.1)
        ctx.md_attach(gp, nve="wall", langevin=False, ucgstate=True)
        ctx.md_setup(nsteps)
        first = ctx.atoms_download()
        ctx.md_run(nsteps, 0)
        gp.check_errors()
        return first, ctx.atoms_download(), ctx.md_info()["nrebuild"] - nre0

    A0, A1, nre = run(25)
    n = beads.n
    assert A0["nlocal"] == n and np.array_equal(np.sort(A0["tag"]), np.arange(1, n + 1))
    f = A0["f"]
    assert np.all(np.isfinite(f)) and np.all(np.isfinite(A0["scores"]))
    assert np.max(np.abs(f.sum(axis=0))) < 1e-9 * np.abs(f).sum() / n * np.sqrt(n) + 1e-7
    assert np.all(A0["ucgforce"] == 0.0)  # table_ucg_bethe never touches ucgforce (UCG/pair_table_ucg_bethe.cpp:457-620)
    # fix ucgstate without a keyword: state = round(ucgp), lambda = ucgp (UCG/fix_ucgstate.cpp:88-132)
    assert np.all((A1["ucgp"] >= 1e-6) & (A1["ucgp"] <= 1 - 1e-6))
    assert np.array_equal(A1["ucgstate"], np.round(A1["ucgp"]).astype(np.int32))
    B0, B1, nre2 = run(25)
    assert nre2 == nre >= 2
    for k in ("x", "v", "f", "ucgl", "ucgp", "scores"):
        assert util.bits_equal(A1[k], B1[k]), k
    assert np.array_equal(A1["tag"], B1["tag"]) and np.array_equal(A1["ucgstate"], B1["ucgstate"])


def test_full_size_properties_4M_beads_density_cluster_switch(fresh_ctx, pkg):
    """BASELINE.json config 5 at its full size on one GPU: 4,000,000 beads (fcc), table_ucg_bethe_density + fix ucgstate mc
    + fix cluster_switch on two actual atom types, molecules of two beads.  Size-independent properties: the molecules
    stay wholly ON or OFF through the switches, switching attempts happen, every bead keeps its identity, the posteriors
    stay inside fix ucgstate's clamp, and a second run from the same input gives the same bits (the Monte-Carlo draws of
    fix ucgstate and the RanPark draws of fix cluster_switch included)."""
    beads = pkg.synth.make_beads(100, seed=12345, lattice="fcc")
    n = beads.n
    assert n == 4_000_000
    deck = util.make_multi_deck(2, "spline", 1024, density=(11.3, 1.5), extra11=0.05, n_file=2000)
    beads.ntypes = 4
    beads.mass = np.array([0.0, 1.0, 1.0, 1.0, 1.0])
    beads.molecule = ((beads.tag - 1) // 2 + 1).astype(np.int32)
    rng = np.random.default_rng(777)
    mtype = rng.integers(1, 3, size=int(beads.molecule.max()) + 1)
    beads.type = mtype[beads.molecule].astype(np.int32)
    wd = util._tmpdir("ucgcs_")
    rates, contacts = pkg.synth.write_cluster_switch_files(wd, 0.35, [1], [2], [(1, 1)])
    mol_seed = int(beads.molecule[np.flatnonzero(beads.type == 1)[0]])
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.002)

    def run(nsteps):
        ctx.upload_beads(beads)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
        gp = util.gpu_pair_multi(ctx, "table_ucg_bethe_density", deck)
        ctx.fix_ucgstate("mc", 9127, 0.01)
        ctx.fix_nve_ucgld_wall_hard(False, 0.1)
        ctx.fix_cluster_switch(mol_seed, 0, 1.2, 4711, 10, rates, contacts)
        ctx.md_attach(gp, nve="wall", langevin=False, ucgstate=True)
        ctx.md_setup(nsteps)
        ctx.md_run(nsteps, 0)
        gp.check_errors()
        return ctx.atoms_download(), ctx.fix_cluster_switch_vector()

    A, va = run(22)
    assert A["nlocal"] == n and np.array_equal(np.sort(A["tag"]), np.arange(1, n + 1))
    # types by tag: both beads of a molecule carry the same type, and some molecules changed theirs
    t = np.empty(n, dtype=np.int32)
    t[A["tag"] - 1] = A["type"]
    assert np.array_equal(t[0::2], t[1::2])
    assert set(np.unique(t)) <= {1, 2}
    assert int((t != beads.type).sum()) > 0
    assert np.all(np.isfinite(A["f"])) and np.all((A["ucgp"] >= 1e-6) & (A["ucgp"] <= 1 - 1e-6))
    B, vb = run(22)
    for k in ("x", "v", "f", "ucgl", "ucgp"):
        assert util.bits_equal(A[k], B[k]), k
    for k in ("tag", "type", "ucgstate"):
        assert np.array_equal(A[k], B[k]), k
    assert np.array_equal(va, vb)
