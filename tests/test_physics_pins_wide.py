"""Reference-independent pins, part 2: what tests/test_physics_pins.py did not cover.  Every check
rebuilds the quantity in numpy / scipy from the formulas of the cited reference lines and from the tables (Pair::single) --
not from the oracle's or the kernels' loops -- and is run on the oracle (CPU tier) and on the HIP kernels (GPU tier).

* table_ucg_bethe_density, all three passes.  The style's forces are (UCG/pair_table_ucg_bethe_density.cpp)
    - the exchange force f = sum_ab p_ab f_ab of every pair (:626-629): the r-derivative of the pair's Bethe free energy
      F_ij = sum_ab p_ab (u_ab + kT ln p_ab) at FIXED priors (the "variational principle" comment :630-631), and
    - the back-propagated collective-variable force (:698-733) cv_i w'(r) of F_i = sum_s p_is mu_s [+ (1 - z_i) kT sum_s p_is ln p_is
      with `entropy`, :302-311], p_i = threshold(rho_i), rho_i = sum_j w(r_ij) (:219-274).
  The pair term's own dependence on the priors is NOT back-propagated by the shipped code: it accumulates
  G[i][0] -= u10 - u00 + kT ln(p10/p00) and G[i][1] -= u11 - u01 + kT ln(p11/p01) (:651-656), multiplies them by dp_i0/drho and
  dp_i1/drho = -dp_i0/drho, and the two brackets are EQUAL at the closure's solution (p11 p00 / (p10 p01) = exp(-J/kT)), so
  the contribution cancels to rounding.  And the back-force carries the sign of the legacy style's code
  (UCG/pair_table_rleucg_interface.cpp:464-487, whose compute_proximity_function_der returns +(1 - tanh^2) / (2 sigma) = -dw/dr
  and is applied along x_i - x_j): it is PLUS the gradient of the one-body terms.  Hence the total force is minus the gradient of
      Phi(x) = sum_pairs F_ij(r_ij; priors frozen) - sum_i F_i(rho_i(x))
  which is rebuilt here from Pair::single, a bisection solution of the closure and tanh, and differentiated by central
  differences (ONE_BODY_SIGN = -1 states that sign; with +1 the test fails by 50 % of the force).  Both are properties of the
  reference's formulas, reproduced, not corrected; DESIGN.md section 2 lists them.  (With the proximity DERIVATIVE in the
  back-force -- the default; the shipped form uses the function itself and is then no gradient at all.)
* table_ucg_bethe `pseudo no`: the full-SCE scores of :583-601 as shipped, evaluated pair by pair in numpy.
* fix ucgld/langevin: gamma1 = -m_lambda / tau / ftm2v, gamma2 = sqrt(m_lambda) / ftm2v * sqrt(24 boltz / tau / dt / mvv2e)
  (UCG/fix_ucgld_langevin.cpp:164-171) and f_lambda += gamma1 v_lambda + gamma2 sqrt(T) (U - 1/2) (:273-291) with U from a
  pure-Python RANMAR (Marsaglia-Zaman-Tsang, seeded as upstream RanMars), in `real` units so that no factor is 1.
* fix cluster_switch: the cluster labels of check_cluster (UCG/fix_cluster_switch.cpp:551-690, mol_offset 0) are the
  smallest molecule id of each connected component of the molecule contact graph: scipy.sparse.csgraph.connected_components."""
import numpy as np
import pytest

import util

H = 1.0e-5
KT = 1.0
ONE_BODY_SIGN = -1.0


# ------------------------------------------------------------------ table_ucg_bethe_density

def _density_case(pkg, entropy):
    deck = util.make_deck("spline", 4096, density=(11.3, 1.5), entropy=entropy, extra11=0.05, n_file=8000)
    b = pkg.synth.make_beads(6, seed=23)  # L = 6.46 > 2 * 2.8: one image per pair inside the list radius
    hp = pkg.capi.Pair(None, "table_ucg_bethe_density")  # host-only: Pair::single as the table evaluator
    hp.settings(deck.pair_style_args())
    hp.coeff(deck.pair_coeff_args())
    hp.init(2, KT)
    return deck, b, hp


def _mindist(b, i):
    d = b.x - b.x[i]
    d -= np.round(d / b.boxhi) * b.boxhi
    return np.sqrt((d * d).sum(axis=1))


def _w(r, r_th=1.5):
    return 0.5 * (1.0 - np.tanh((r - r_th) / (0.1 * r_th)))  # compute_proximity_function, :118-121


def _prior1(rho, rho_th=11.3):
    return 1.0 - (0.5 + 0.5 * np.tanh((rho - rho_th) / (0.1 * rho_th)))  # p_i1 = 1 - p_i0, :107-113 and :251-254


def _rho(b, i):
    r = _mindist(b, i)
    sel = (r < 2.5) & (np.arange(b.n) != i)
    return float(_w(r[sel]).sum())


def _closure_p(pi1, pj1, J):
    # stationarity p11 p00 = exp(-J/kT) p10 p01, by bisection on p11 (not the style's closed form)
    lo, hi = max(0.0, pi1 + pj1 - 1.0) + 1e-15, min(pi1, pj1) - 1e-15
    g = lambda p: np.log(p) + np.log(1 + p - pi1 - pj1) - np.log(pi1 - p) - np.log(pj1 - p) + J / KT  # noqa: E731
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if g(mid) > 0:
            hi = mid
        else:
            lo = mid
    p11 = 0.5 * (lo + hi)
    return np.array([[1 + p11 - pi1 - pj1, pj1 - p11], [pi1 - p11, p11]])


def _phi_of_bead(b, hp, i, prior1_frozen, z_frozen, entropy, mu=(0.0, 0.5)):
    """the terms of Phi that change when bead i moves: its pairs' Bethe free energies at the FROZEN priors, and the one-body
    terms of i and of every bead whose local density i contributes to"""
    r = _mindist(b, i)
    nb = np.flatnonzero((r < 2.5) & (np.arange(b.n) != i))
    F = 0.0
    for m in nb:
        u = np.array([[hp.single(1 + a, 1 + c, r[m] * r[m])[0] for c in (0, 1)] for a in (0, 1)])
        p = _closure_p(prior1_frozen[i], prior1_frozen[m], u[1, 1] + u[0, 0] - u[0, 1] - u[1, 0])
        F += float(np.sum(p * (u + KT * np.log(p))))
    for k in [i, *nb]:
        p1 = _prior1(_rho(b, k))
        ps = np.array([1.0 - p1, p1])
        F += ONE_BODY_SIGN * float(np.dot(ps, mu))
        if entropy:
            F += ONE_BODY_SIGN * (1.0 - z_frozen[k]) * KT * float(np.sum(ps * np.log(ps)))
    return F


def _density_fd(pkg, b, hp, f0, entropy, picks):
    prior1 = np.array([_prior1(_rho(b, k)) for k in range(b.n)])
    z = np.array([int(((_mindist(b, k) < 2.8).sum()) - 1) for k in range(b.n)])  # jnum: the whole row, skin included (:296)
    worst = 0.0
    used = 0
    for i, d in picks:
        r = _mindist(b, i)
        r[i] = 0.0
        if np.min(np.abs(r - 2.5)) < 50 * H:  # cut tables: Phi jumps when a pair crosses the cutoff
            continue
        bp, bm = _copy(pkg, b), _copy(pkg, b)
        bp.x[i, d] += H
        bm.x[i, d] -= H
        g = -(_phi_of_bead(bp, hp, i, prior1, z, entropy) - _phi_of_bead(bm, hp, i, prior1, z, entropy)) / (2 * H)
        worst = max(worst, abs(g - f0[i, d]) / max(1.0, abs(f0[i, d])))
        used += 1
    assert used >= 3
    return worst, prior1


def _copy(pkg, b):
    c = pkg.synth.make_beads(5, seed=1)
    for k in ("x", "v", "type", "tag", "mask", "ucgstate", "ucgl", "ucgvl", "ucgml", "ucgp"):
        setattr(c, k, np.array(getattr(b, k), copy=True))
    c.n, c.boxlo, c.boxhi, c.mass, c.ntypes = b.n, b.boxlo.copy(), b.boxhi.copy(), b.mass.copy(), b.ntypes
    return c


PICKS = [(0, 0), (17, 1), (63, 2), (101, 0), (124, 1), (150, 2), (200, 0)]


def _oracle_density_forces(pkg, b, deck):
    op = util.oracle_pair("table_ucg_bethe_density", deck, T=KT)
    sim = util.oracle_sim(b, op, mode=1)
    sim.rebuild()
    assert sim.compute_forces(1, 1) == 0
    A = sim.arrays()
    o = np.argsort(A["tag"])
    return A["f"][o], A["ucgp"][o]


def _gpu_density_forces(ctx, pkg, b, deck):
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.upload_beads(b)
    ctx.domain_set(b.boxlo, b.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    ctx.neigh_rebuild()
    gp = util.gpu_pair(ctx, "table_ucg_bethe_density", deck, T=KT)
    gp.compute(1, 1)
    gp.check_errors()
    A = ctx.atoms_download()
    o = np.argsort(A["tag"])
    gp.close()
    return A["f"][o], A["ucgp"][o]


@pytest.mark.parametrize("entropy", [False, True])
def test_oracle_density_style_forces_are_minus_the_gradient_of_its_functional(pkg, orc, entropy):
    deck, b, hp = _density_case(pkg, entropy)
    f0, _ = _oracle_density_forces(pkg, b, deck)
    worst, prior1 = _density_fd(pkg, b, hp, f0, entropy, PICKS)
    assert prior1.std() > 0.05  # a mixed population: the threshold function is exercised on its slope
    assert worst < 2e-6, worst
    assert abs(f0.sum(axis=0)).max() < 1e-9  # every neighbour pushes back: momentum is conserved


@pytest.mark.gpu
@pytest.mark.parametrize("entropy", [False, True])
def test_gpu_density_style_forces_are_minus_the_gradient_of_its_functional(fresh_ctx, pkg, entropy):
    deck, b, hp = _density_case(pkg, entropy)
    f0, _ = _gpu_density_forces(fresh_ctx, pkg, b, deck)
    worst, _ = _density_fd(pkg, b, hp, f0, entropy, PICKS)
    assert worst < 2e-6, worst


# ------------------------------------------------------------------ table_ucg_bethe, full-SCE scores

def _sce_case(pkg):
    deck = util.make_deck("spline", 1024, extra_keywords=("method", "bethe", "pseudo", "no", "prior", "ucgl"))
    b = pkg.synth.make_beads(5, seed=41)
    rng = np.random.default_rng(7)
    b.ucgl = np.clip(b.ucgl, 0.05, 0.95)
    b.ucgp = np.clip(rng.uniform(size=b.n), 0.05, 0.95)
    hp = pkg.capi.Pair(None, "table_ucg_bethe")
    hp.settings(deck.pair_style_args())
    hp.coeff(deck.pair_coeff_args())
    hp.init(2, KT)
    return deck, b, hp


def _sce_scores_numpy(b, hp, mu=(0.0, 0.5)):
    """UCG/pair_table_ucg_bethe.cpp:155-162 (scores start from -mu_s / kT) and :583-601 AS SHIPPED:
         S[i][0] -= (p00/pi0 u00 + p10/pi1 u01) / kT      S[i][1] -= (p01/pi0 u10 + p11/pi1 u11) / kT
         S[j][0] -= (p00/pj0 u00 + p10/pj0 u01) / kT      S[j][1] -= (p01/pj1 u10 + p11/pj1 u11) / kT
    (p_ab: i in state a, j in state b).  The textbook conditionals p(sj|si) would put p01/pi0 next to u01 and p10/pi1 next to
    u10 in the first line; `textbook` returns that variant so that the test can show the two differ."""
    S = np.tile(-np.asarray(mu) / KT, (b.n, 1))
    T = S.copy()
    for i in range(b.n):
        d = b.x - b.x[i]
        d -= np.round(d / b.boxhi) * b.boxhi
        rsq = (d * d).sum(axis=1)
        for m in np.flatnonzero((rsq < 6.25) & (b.tag > b.tag[i])):  # every pair once, i = the lower tag
            u = np.array([[hp.single(1 + a, 1 + c, rsq[m])[0] for c in (0, 1)] for a in (0, 1)])
            pi1, pj1 = b.ucgl[i], b.ucgp[m]  # priors after the first call: "i" from ucgl, "j" from ucgp (:199-205, :247-253)
            pi0, pj0 = 1.0 - pi1, 1.0 - pj1
            p = _closure_p(pi1, pj1, u[1, 1] + u[0, 0] - u[0, 1] - u[1, 0])
            S[i, 0] -= (p[0, 0] / pi0 * u[0, 0] + p[1, 0] / pi1 * u[0, 1]) / KT
            S[i, 1] -= (p[0, 1] / pi0 * u[1, 0] + p[1, 1] / pi1 * u[1, 1]) / KT
            S[m, 0] -= (p[0, 0] / pj0 * u[0, 0] + p[1, 0] / pj0 * u[0, 1]) / KT
            S[m, 1] -= (p[0, 1] / pj1 * u[1, 0] + p[1, 1] / pj1 * u[1, 1]) / KT
            T[i, 0] -= (p[0, 0] / pi0 * u[0, 0] + p[0, 1] / pi0 * u[0, 1]) / KT
            T[i, 1] -= (p[1, 0] / pi1 * u[1, 0] + p[1, 1] / pi1 * u[1, 1]) / KT
    return S, T


def _scores(A):
    o = np.argsort(A["tag"])
    return A["scores"][o]


def test_oracle_full_sce_scores_equal_the_shipped_formulas_evaluated_in_numpy(pkg, orc):
    deck, b, hp = _sce_case(pkg)
    S, T = _sce_scores_numpy(b, hp)
    for mode in (0, 1):
        op = util.oracle_pair("table_ucg_bethe", deck, T=KT)
        sim = util.oracle_sim(b, op, mode=mode)
        sim.rebuild()
        assert sim.compute_forces(0, 0) == 0
        got = _scores(sim.arrays())
        assert np.max(np.abs(got - S)) < 1e-9 * np.max(np.abs(S)), mode
    # the shipped first line is not the textbook conditional expectation: documented, not corrected
    assert np.max(np.abs(S[:, 0] - T[:, 0])) > 1e-3


@pytest.mark.gpu
def test_gpu_full_sce_scores_equal_the_shipped_formulas_evaluated_in_numpy(fresh_ctx, pkg):
    deck, b, hp = _sce_case(pkg)
    S, _ = _sce_scores_numpy(b, hp)
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.upload_beads(b)
    ctx.domain_set(b.boxlo, b.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    ctx.neigh_rebuild()
    gp = util.gpu_pair(ctx, "table_ucg_bethe", deck, T=KT)
    gp.compute(0, 0)
    gp.check_errors()
    got = _scores(ctx.atoms_download())
    gp.close()
    assert np.max(np.abs(got - S)) < 1e-9 * np.max(np.abs(S))


# ------------------------------------------------------------------ fix ucgld/langevin prefactors

class PyRanMars:
    """RANMAR (Marsaglia, Zaman, Tsang 1990) seeded as upstream LAMMPS' RanMars, plain Python"""

    def __init__(self, seed):
        ij = (seed - 1) // 30082
        kl = (seed - 1) - 30082 * ij
        i, j, k, l = (ij // 177) % 177 + 2, ij % 177 + 2, (kl // 169) % 178 + 1, kl % 169
        self.u = [0.0] * 98
        for ii in range(1, 98):
            s, t = 0.0, 0.5
            for _ in range(24):
                m = ((i * j) % 179) * k % 179
                i, j, k = j, k, m
                l = (53 * l + 1) % 169
                if (l * m) % 64 >= 32:
                    s += t
                t *= 0.5
            self.u[ii] = s
        self.c, self.cd, self.cm = 362436.0 / 16777216.0, 7654321.0 / 16777216.0, 16777213.0 / 16777216.0
        self.i97, self.j97 = 97, 33
        self.uniform()

    def uniform(self):
        uni = self.u[self.i97] - self.u[self.j97]
        if uni < 0.0:
            uni += 1.0
        self.u[self.i97] = uni
        self.i97 -= 1
        if self.i97 == 0:
            self.i97 = 97
        self.j97 -= 1
        if self.j97 == 0:
            self.j97 = 97
        self.c -= self.cd
        if self.c < 0.0:
            self.c += self.cm
        uni -= self.c
        if uni < 0.0:
            uni += 1.0
        return uni


REAL = dict(boltz=0.0019872067, ftm2v=1.0 / 48.88821291 / 48.88821291, mvv2e=48.88821291 * 48.88821291)


def _langevin_expected(b, ml, tau, dt, T, seed, steps):
    g1 = -ml / tau / REAL["ftm2v"]
    g2 = np.sqrt(ml) / REAL["ftm2v"] * np.sqrt(24.0 * REAL["boltz"] / tau / dt / REAL["mvv2e"])
    rng = PyRanMars(seed)
    out = []
    for _ in range(steps):
        U = np.array([rng.uniform() for _ in range(b.n)])  # one draw per owned atom of the group, in index order (:273-291)
        out.append(g1 * b.ucgvl + g2 * np.sqrt(T) * (U - 0.5))
    return out


def _langevin_case(pkg):
    b = pkg.synth.make_beads(4, seed=3)
    b.ucgvl = np.random.default_rng(11).normal(size=b.n) * 0.01
    b.ucgml[:] = 7.5  # all equal: init() reads ucgml[type index]
    return b


def test_oracle_langevin_prefactors_equal_the_closed_forms(pkg, orc):
    b = _langevin_case(pkg)
    tau, dt, T, seed = 120.0, 2.0, 310.0, 4711
    want = _langevin_expected(b, 7.5, tau, dt, T, seed, 2)
    L = orc.lib()
    sim = orc.Sim(b)  # only as the owner of an orc_atoms record holding the beads
    atoms = L.orc_sim_atoms(sim.h)
    fx = L.orc_fix_langevin_create(2, T, T, tau, seed, 0)
    try:
        L.orc_fix_langevin_init(fx, atoms, dt, REAL["boltz"], REAL["ftm2v"], REAL["mvv2e"])
        for k in range(2):
            before = sim.arrays()["ucgforce"].copy()
            L.orc_fix_langevin_post_force(fx, atoms, 1, k, 0, 10)
            got = sim.arrays()["ucgforce"] - before
            assert np.allclose(got, want[k], rtol=1e-12, atol=0.0), k
            assert np.abs(got).max() > 0.0
    finally:
        L.orc_fix_langevin_destroy(fx)


@pytest.mark.gpu
def test_gpu_langevin_prefactors_equal_the_closed_forms(pkg):
    b = _langevin_case(pkg)
    tau, dt, T, seed = 120.0, 2.0, 310.0, 4711
    want = _langevin_expected(b, 7.5, tau, dt, T, seed, 3)
    ctx = pkg.capi.Context(-1, dt=dt, **REAL)
    try:
        ctx.upload_beads(b)
        ctx.fix_ucgld_langevin(T, T, tau, seed)
        ctx.fix_ucgld_langevin_init(2, b.ucgml[:3])
        for k in range(3):
            ctx.force_clear()
            ctx.fix_ucgld_langevin_post_force(k, 0, 10)
            got = ctx.atoms_download()["ucgforce"]
            assert np.allclose(got, want[k], rtol=1e-12, atol=0.0), k
            assert np.abs(got).max() > 0.0
        # reset_target (:358-361): sqrt(T) follows; reset_dt (:366-376) AS SHIPPED: gamma2 from atom->mass, gamma1 untouched
        ctx.fix_ucgld_langevin_reset_target(2.0 * T)
        ctx.force_clear()
        ctx.fix_ucgld_langevin_post_force(3, 0, 10)
        rng = PyRanMars(seed)
        for _ in range(3 * b.n):
            rng.uniform()
        U = np.array([rng.uniform() for _ in range(b.n)])
        g1 = -7.5 / tau / REAL["ftm2v"]
        g2 = np.sqrt(7.5) / REAL["ftm2v"] * np.sqrt(24.0 * REAL["boltz"] / tau / dt / REAL["mvv2e"])
        assert np.allclose(ctx.atoms_download()["ucgforce"], g1 * b.ucgvl + g2 * np.sqrt(2.0 * T) * (U - 0.5), rtol=1e-12, atol=0.0)
        ctx.set_units(REAL["boltz"], REAL["ftm2v"], REAL["mvv2e"], 0.5 * dt)
        ctx.fix_ucgld_langevin_reset_dt(2, b.mass)
        ctx.force_clear()
        ctx.fix_ucgld_langevin_post_force(4, 0, 10)
        U = np.array([rng.uniform() for _ in range(b.n)])
        g2m = np.sqrt(b.mass[b.type]) / REAL["ftm2v"] * np.sqrt(24.0 * REAL["boltz"] / tau / (0.5 * dt) / REAL["mvv2e"])
        assert np.allclose(ctx.atoms_download()["ucgforce"], g1 * b.ucgvl + g2m * np.sqrt(2.0 * T) * (U - 0.5), rtol=1e-12, atol=0.0)
        # fix_modify temp with a bias-removing compute: post_force_templated<1> zeroes the random force where v_lambda == 0
        vl = b.ucgvl.copy()
        vl[::3] = 0.0
        ctx.atoms_upload_owned(ucgvl=vl)
        ctx.fix_ucgld_langevin_set_bias(True)
        ctx.force_clear()
        ctx.fix_ucgld_langevin_post_force(5, 0, 10)
        U = np.array([rng.uniform() for _ in range(b.n)])
        fran = g2m * np.sqrt(2.0 * T) * (U - 0.5)
        fran[::3] = 0.0
        assert np.allclose(ctx.atoms_download()["ucgforce"], g1 * vl + fran, rtol=1e-12, atol=0.0)
    finally:
        ctx.close()


# ------------------------------------------------------------------ fix cluster_switch labels

def _contact_components(b, cutoff, contact_types=((1, 1),)):
    """molecules are nodes; an edge where two beads of different molecules with an allowed (type_i, type_j) pair lie inside
    the cutoff (minimum image); the label of a molecule = the smallest molecule id of its component"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    nm = int(b.molecule.max()) + 1
    rows, cols = [], []
    allowed = set(contact_types)
    for i in range(b.n):
        d = b.x - b.x[i]
        d -= np.round(d / b.boxhi) * b.boxhi
        rsq = (d * d).sum(axis=1)
        for m in np.flatnonzero((rsq < cutoff * cutoff) & (b.molecule != b.molecule[i])):
            if (int(b.type[i]), int(b.type[m])) in allowed:
                rows.append(int(b.molecule[i]))
                cols.append(int(b.molecule[m]))
    g = coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(nm, nm))
    _, comp = connected_components(g, directed=False)
    label = np.full(nm, -1)
    present = np.unique(b.molecule)
    for c in np.unique(comp[present]):
        members = present[comp[present] == c]
        label[members] = members.min()
    return label


def _cluster_case(pkg):
    deck = util.make_multi_deck(2, "spline", 256)
    b = util.multi_type_beads(pkg, 10, 2, seed=5, molecule_size=2)
    rates, contacts = pkg.synth.write_cluster_switch_files(deck.workdir, 0.35, [1], [2], [(1, 1)])
    mol_seed = int(b.molecule[np.flatnonzero(b.type == 1)[0]])
    return deck, b, rates, contacts, mol_seed


def test_oracle_cluster_labels_are_the_connected_components_of_the_contact_graph(pkg, orc):
    deck, b, rates, contacts, mol_seed = _cluster_case(pkg)
    cutoff = 1.15
    want = _contact_components(b, cutoff)
    op = util.oracle_pair_multi("table_ucgld", deck)
    sim = util.oracle_sim(b, op, mode=1)
    sim.rebuild()
    sim.cluster_switch(mol_seed, 0, cutoff, 4711, 5, rates, contacts)
    L = orc.lib()
    cs = L.orc_sim_cs(sim.h)
    assert L.orc_cs_check_cluster(cs, L.orc_sim_atoms(sim.h), L.orc_sim_molecule(sim.h), L.orc_sim_full_list(sim.h)) == 0
    got = sim.cs_arrays()["mol_cluster"]
    present = np.unique(b.molecule)
    assert np.array_equal(got[present], want[present])
    sizes = np.bincount(want[present])
    assert sizes.max() > 3 and (sizes == 1).sum() > 3  # real clusters and isolated molecules


@pytest.mark.gpu
def test_gpu_cluster_labels_are_the_connected_components_of_the_contact_graph(fresh_ctx, pkg):
    deck, b, rates, contacts, mol_seed = _cluster_case(pkg)
    cutoff = 1.15
    want = _contact_components(b, cutoff)
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.upload_beads(b)
    ctx.domain_set(b.boxlo, b.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    ctx.neigh_rebuild()
    ctx.fix_cluster_switch(mol_seed, 0, cutoff, 4711, 5, rates, contacts)
    ctx.fix_cluster_switch_check_cluster()
    got = ctx.fix_cluster_switch_arrays()["mol_cluster"]
    present = np.unique(b.molecule)
    assert np.array_equal(got[present], want[present])
