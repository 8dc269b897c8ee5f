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
