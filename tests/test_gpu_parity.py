"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle on the same inputs.

Bar: bit-exact against the oracle's canonical-order mode for every per-bead output
(f, ucgforce, ucgsoftmaxscores, ucgp, states, RNG streams); <= 1e-11 relative against the
oracle's reference-order mode (half list + scatter, the reference's own summation order),
which differs from the canonical order only by floating-point re-association.
"""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

REL_REFORDER = 1e-11


def test_library_loads_and_device(gpu_ctx, pkg):
    assert pkg.capi.lib().ucg_abi_version() == 1
    nl, ng = gpu_ctx.counts()
    assert nl >= 0 and ng >= 0


@pytest.mark.parametrize("seed,skip,n", [(48279, 0, 10000), (12345, 99, 70000), (900000000, 1234567, 5000), (1, 0, 4097)])
def test_ranmars_device_stream(gpu_ctx, orc, seed, skip, n):
    L = orc.lib()
    r = orc.RanMars()
    L.orc_ranmars_init(r, seed)
    ref = np.zeros(skip + n)
    L.orc_ranmars_fill(r, skip + n, ref.ctypes.data_as(orc.c_double_p))
    got = gpu_ctx.ranmars_fill(seed, skip, n)
    assert util.bits_equal(got, ref[skip:])


def test_ranmars_published_check_values(gpu_ctx):
    # Marsaglia, Zaman & Tsang: seeds ij=1802, kl=9373 -> after 20000 draws the next six * 2^24
    got = gpu_ctx.ranmars_fill(1802 * 30082 + 9373 + 1, 19999, 6)
    assert [int(v * 16777216.0) for v in got] == [6533892, 14220222, 7275067, 6172232, 8354498, 10633180]


@pytest.mark.parametrize("tabstyle,tablength", [("spline", 1024), ("linear", 4096), ("lookup", 2000), ("spline", 8000),
                                                ("bitmap", 12), ("bitmap", 8)])
@pytest.mark.parametrize("ncell", [6, 12])
def test_pair_ucgld_parity(gpu_ctx, pkg, orc, tabstyle, tablength, ncell):
    deck = util.make_deck(tabstyle, tablength)
    beads = pkg.synth.make_beads(ncell, seed=100 + ncell)
    op = util.oracle_pair("table_ucgld", deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    util.upload_from_oracle(gpu_ctx, sim, beads)
    gp = util.gpu_pair(gpu_ctx, "table_ucgld", deck)
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = gpu_ctx.atoms_download()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    for k in ("f", "ucgforce", "scores", "num_ucgstates"):
        assert np.array_equal(G[k], O[k]), k
    assert util.bits_equal(G["f"], O["f"]) and util.bits_equal(G["scores"], O["scores"])
    ev = sim.ev()
    assert abs(eng - ev["eng_vdwl"]) <= 1e-12 * abs(ev["eng_vdwl"])
    assert np.allclose(vir, ev["virial"], rtol=1e-11, atol=1e-9)
    # reference order (half list, scatter, reverse sum): same numbers, different association
    sim0 = util.oracle_sim(beads, op, mode=0)
    sim0.rebuild()
    assert sim0.compute_forces(1, 1) == 0
    R = sim0.arrays()
    scale = np.abs(R["f"]).max()
    assert np.abs(G["f"] - R["f"]).max() <= REL_REFORDER * scale
    assert np.abs(G["ucgforce"] - R["ucgforce"]).max() <= REL_REFORDER * np.abs(R["ucgforce"]).max()
    assert np.abs(G["scores"] - R["scores"]).max() <= REL_REFORDER * np.abs(R["scores"]).max()
    assert abs(eng - sim0.ev()["eng_vdwl"]) <= 1e-11 * abs(eng)


@pytest.mark.parametrize("extra", [(), ("pseudo", "no"), ("method", "mf"), ("prior", "chemical_potential")])
@pytest.mark.parametrize("first_call", [True, False])
def test_pair_bethe_parity(gpu_ctx, pkg, orc, extra, first_call):
    deck = util.make_deck("spline", 1024, extra_keywords=extra)
    beads = pkg.synth.make_beads(8, seed=77)
    if not first_call:
        rng = np.random.default_rng(5)
        beads.ucgp = np.clip(rng.uniform(size=beads.n), 1e-6, 1 - 1e-6)
    op = util.oracle_pair("table_ucg_bethe", deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    util.upload_from_oracle(gpu_ctx, sim, beads)
    gp = util.gpu_pair(gpu_ctx, "table_ucg_bethe", deck)
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = gpu_ctx.atoms_download()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    assert util.bits_equal(G["f"], O["f"])
    assert util.bits_equal(G["scores"], O["scores"])
    assert np.array_equal(G["ucgforce"], np.zeros_like(G["ucgforce"]))
    ev = sim.ev()
    assert abs(eng - ev["eng_vdwl"]) <= 1e-12 * abs(ev["eng_vdwl"])
    sim0 = util.oracle_sim(beads, op, mode=0)
    sim0.rebuild()
    assert sim0.compute_forces(1, 1) == 0
    R = sim0.arrays()
    assert np.abs(G["f"] - R["f"]).max() <= REL_REFORDER * np.abs(R["f"]).max()
    assert np.abs(G["scores"] - R["scores"]).max() <= REL_REFORDER * np.abs(R["scores"]).max()


def test_table_range_error_is_reported(gpu_ctx, pkg, orc):
    deck = util.make_deck("spline", 512)
    beads = pkg.synth.make_beads(6, seed=3)
    beads.x[1] = beads.x[0] + np.array([0.3, 0.0, 0.0])  # closer than the table's inner cutoff 0.6
    op = util.oracle_pair("table_ucgld", deck)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    util.upload_from_oracle(gpu_ctx, sim, beads)
    gp = util.gpu_pair(gpu_ctx, "table_ucgld", deck)
    gp.compute(0, 0)
    with pytest.raises(pkg.capi.UcgError) as ei:
        gp.check_errors()
    assert "inner cutoff" in str(ei.value)
    assert sim.compute_forces(0, 0) != 0  # the oracle flags the same pair


@pytest.mark.parametrize("mode", ["ld", None, "mc"])
def test_fix_hooks_parity(gpu_ctx, pkg, orc, mode):
    """one Verlet step assembled from the individual hooks, compared hook by hook"""
    L = orc.lib()
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(8, seed=11)
    dt = 0.002
    op = util.oracle_pair("table_ucgld", deck)
    sim = util.oracle_sim(beads, op, mode=1, dt=dt)
    sim.rebuild()
    gpu_ctx.set_units(1.0, 1.0, 1.0, dt)
    util.upload_from_oracle(gpu_ctx, sim, beads)
    gp = util.gpu_pair(gpu_ctx, "table_ucgld", deck)
    # --- pair
    gp.compute(0, 0)
    assert sim.compute_forces(0, 0) == 0
    a = L.orc_sim_atoms(sim.h)
    # --- langevin
    lang = L.orc_fix_langevin_create(2, 1.0, 1.5, 0.7, 48279, 0)
    L.orc_fix_langevin_init(lang, a, dt, 1.0, 1.0, 1.0)
    O0 = sim.arrays()
    gpu_ctx.fix_ucgld_langevin(1.0, 1.5, 0.7, 48279)
    gpu_ctx.fix_ucgld_langevin_init(2, O0["ucgml"][:3])
    for step in (3, 4):
        L.orc_fix_langevin_post_force(lang, a, 1, step, 0, 10)
        gpu_ctx.fix_ucgld_langevin_post_force(step, 0, 10)
    assert abs(gpu_ctx.fix_ucgld_langevin_t_target() - (1.0 + 0.4 * 0.5)) < 1e-15
    G = gpu_ctx.atoms_download()
    O = sim.arrays()
    assert util.bits_equal(G["ucgforce"], O["ucgforce"])
    # --- ucgstate
    fx = L.orc_fix_ucgstate_create(1 if mode == "ld" else 0, 1 if mode == "mc" else 0, 9127, 0.3, 0)
    gpu_ctx.fix_ucgstate(mode, 9127, 0.3)
    for _ in range(2):
        L.orc_fix_ucgstate_post_force(fx, a)
        gpu_ctx.fix_ucgstate_post_force()
    G = gpu_ctx.atoms_download()
    O = sim.arrays()
    assert util.bits_equal(G["ucgp"], O["ucgp"])
    assert np.array_equal(G["ucgstate"], O["ucgstate"])
    assert util.bits_equal(G["ucgl"], O["ucgl"])
    if mode == "mc":
        assert 0 < G["ucgstate"].sum() < beads.n
    # --- nve
    L.orc_fix_nve_final(a, dt, 1.0, 1)
    gpu_ctx.fix_nve_ucgld_final_integrate()
    L.orc_fix_nve_initial(a, dt, 1.0, 1)
    gpu_ctx.fix_nve_ucgld_initial_integrate()
    G = gpu_ctx.atoms_download()
    O = sim.arrays()
    for k in ("x", "v", "ucgl", "ucgvl"):
        assert util.bits_equal(G[k], O[k]), k
    L.orc_fix_langevin_end_of_step(lang, a, 1, 1.0, 1.0)
    out = np.zeros(16)
    L.orc_fix_langevin_get(lang, out.ctypes.data_as(orc.c_double_p))
    lt = gpu_ctx.fix_ucgld_langevin_end_of_step()
    assert abs(lt - out[2]) <= 1e-13 * abs(out[2])
    L.orc_fix_langevin_destroy(lang)
    L.orc_fix_ucgstate_destroy(fx)


@pytest.mark.parametrize("barrier", [0.1, 0.37])
def test_fix_nve_ucgld_wall_hard_hooks_bitwise(gpu_ctx, pkg, orc, barrier):
    """post_force (bias) -> final_integrate (reflection) -> initial_integrate (state from lambda), hook by hook"""
    L = orc.lib()
    dt = 0.004
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(7, seed=77)
    rng = np.random.default_rng(5)
    beads.ucgl = rng.uniform(-0.4, 1.4, beads.n)  # a third of the beads sit beyond a wall
    beads.ucgvl = rng.normal(0.0, 3.0, beads.n)
    op = util.oracle_pair("table_ucgld", deck)
    sim = util.oracle_sim(beads, op, mode=1, dt=dt)
    sim.rebuild()
    assert sim.compute_forces(0, 0) == 0
    gpu_ctx.set_units(1.0, 1.0, 1.0, dt)
    util.upload_from_oracle(gpu_ctx, sim, beads)
    gp = util.gpu_pair(gpu_ctx, "table_ucgld", deck)
    gp.compute(0, 0)
    a = L.orc_sim_atoms(sim.h)
    gpu_ctx.fix_nve_ucgld_wall_hard(True, barrier)
    for _ in range(3):
        L.orc_fix_nve_wall_post_force(a, barrier, 1)
        gpu_ctx.fix_nve_ucgld_wall_hard_post_force()
        assert util.bits_equal(gpu_ctx.atoms_download()["ucgforce"], sim.arrays()["ucgforce"])
        L.orc_fix_nve_wall_final(a, dt, 1.0, 1)
        gpu_ctx.fix_nve_ucgld_wall_hard_final_integrate()
        L.orc_fix_nve_wall_initial(a, dt, 1.0, 1)
        gpu_ctx.fix_nve_ucgld_wall_hard_initial_integrate()
        G, O = gpu_ctx.atoms_download(), sim.arrays()
        for k in ("x", "v", "ucgl", "ucgvl"):
            assert util.bits_equal(G[k], O[k]), k
        assert np.array_equal(G["ucgstate"], O["ucgstate"])
        assert np.array_equal(G["ucgstate"] == 1, G["ucgl"] >= 0.5)
    gpu_ctx.fix_nve_ucgld_wall_hard(False, 0.1)


@pytest.mark.parametrize("kT", [1.0, 0.7, 2.4942, 1e-3, 300.0 * 0.0019872067])
def test_exact_division_by_constant(gpu_ctx, kT):
    # the FAST kernels' u / kT: reciprocal + two FMA corrections must equal the IEEE quotient
    assert gpu_ctx.selftest_div(kT, 12345, 4_000_000) == 0


def test_bare_division_core_equals_the_ieee_division_on_its_operand_ranges(gpu_ctx):
    """csrc/ucg_math.h: ucg_div_core -- the hardware division's Newton-Raphson core without v_div_scale / v_div_fixup -- is
    used for the two quotients of exp / expm1's common path, whose operands lie where the scaling is the identity: 8 M random
    operand pairs of those ranges (and zero numerators), every quotient bit-equal to a / b"""
    for seed in (1, 987654321):
        assert gpu_ctx.selftest_div_core(seed, 4_000_000) == 0


def test_sqrt_core_matches_the_hardware_square_root(gpu_ctx):
    """ucg_sqrt_core (csrc/ucg_math.h): the rsq-seeded iteration of the hardware square root without its operand scaling and
    special-case selects, used by the Bethe closure for arguments in [2^-700, 2^700) where those are the identity: 8 M
    random arguments of that range, one in eight a perfect square or its neighbour in the last place, every root bit-equal
    to sqrt(x)"""
    for seed in (3, 123456789):
        assert gpu_ctx.selftest_sqrt_core(seed, 4_000_000) == 0


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
@pytest.mark.parametrize("tabstyle,tablength,T", [("spline", 1024, 0.7), ("linear", 2048, 1.3), ("lookup", 3000, 1.0), ("spline", 512, 0.25),
                                                  ("bitmap", 10, 0.7)])
def test_fast_and_generic_kernels_give_the_same_bits(fresh_ctx, pkg, orc, style, tabstyle, tablength, T):
    """the full-row gather kernels' fast variants (one shared r^2 grid, exact reciprocal divisions) against their
    general code path: same bits, both equal to the oracle's ordered sums (option pair_vrow 0: the virtual-row kernels
    have no general variant, and sum differently)"""
    ctx = fresh_ctx
    ctx.set_option("pair_vrow", 0)
    deck = util.make_deck(tabstyle, tablength)
    beads = pkg.synth.make_beads(9, seed=55)
    beads.ucgp = np.clip(np.random.default_rng(2).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    op = util.oracle_pair(style, deck, T=T)
    op.set_sum_fixed(False)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    res = {}
    for generic in (0, 1):
        ctx.set_option("generic_kernels", generic)
        util.upload_from_oracle(ctx, sim, beads)
        gp = util.gpu_pair(ctx, style, deck, T=T)
        assert not gp.sum_fixed
        res[generic] = (gp.compute(1, 1), ctx.atoms_download())
        gp.check_errors()
        if generic == 0:
            # the variants that read their rows with non-temporal loads (chosen by the list's size; forced here): the same bits
            ctx.set_option("stream_rows", 1)
            gp.compute(0, 0)
            gp.check_errors()
            S = ctx.atoms_download()
            ctx.set_option("stream_rows", -1)
            assert util.bits_equal(S["f"], O["f"]) and util.bits_equal(S["scores"], O["scores"])
        gp.close()
    for generic in (0, 1):
        G = res[generic][1]
        assert util.bits_equal(G["f"], O["f"]) and util.bits_equal(G["scores"], O["scores"]), generic
        assert util.bits_equal(G["ucgforce"], O["ucgforce"]), generic
    assert res[0][0][0] == res[1][0][0]


def test_generic_path_nonuniform_grids_and_special_lj(fresh_ctx, pkg, orc):
    # table 11 starts at a different inner radius (its own r^2 grid), and special_lj != 1 with
    # flagged entries: exercises the per-table lookup and factor_lj of the general kernels
    ctx = fresh_ctx
    deck = util.make_deck("spline", 700, rlo11=0.7)
    beads = pkg.synth.make_beads(8, seed=8)
    op = util.oracle_pair("table_ucgld", deck)
    L = orc.lib()
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    il, nn, fi, ne = sim.full_list()
    ne = ne.copy()
    rng = np.random.default_rng(1)
    flag = rng.integers(0, 4, size=ne.size).astype(np.int64)
    ne = (ne.astype(np.int64) | (flag << 30)).astype(np.uint32).view(np.int32)
    special = (1.0, 0.0, 0.5, 0.25)
    # oracle on the flagged list
    A = sim.arrays(ghosts=True)
    arr = orc.AtomArrays(A["nlocal"], A["nghost"])
    for k in ("x", "type", "tag", "ucgstate", "ucgl", "ucgp"):
        getattr(arr, k)[...] = A[k]
    lst, keep = orc.list_from_csr(il, nn, fi, ne)
    import ctypes as C
    sl = (C.c_double * 4)(*special)
    L.orc_pair_set_special_lj.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.orc_pair_set_special_lj(op.h, sl)
    ev = orc.Ev()
    cs = arr.cstruct()
    assert L.orc_pair_compute_gather(op.h, C.byref(cs), C.byref(lst), 1, 1, C.byref(ev)) == 0
    ctx.set_units(1.0, 1.0, 1.0, 0.002, special)
    nl, ng = A["nlocal"], A["nghost"]
    ctx.atoms_upload(nl, ng, 2, A["x"], A["v"], A["type"], A["tag"], A["mask"][:nl], A["ucgstate"], A["ucgl"],
                     A["ucgvl"], A["ucgml"], A["ucgp"], beads.mass)
    ctx.neigh_upload_full(nn, fi, ne)
    gp = util.gpu_pair(ctx, "table_ucgld", deck)
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    assert util.bits_equal(G["f"], arr.f[:nl]) and util.bits_equal(G["scores"], arr.scores[:nl])
    assert util.bits_equal(G["ucgforce"], arr.ucgforce[:nl])
    assert abs(eng - ev.eng_vdwl) <= 1e-12 * abs(ev.eng_vdwl)


DENS = dict(density=(11.3, 1.5), extra11=0.05)


@pytest.mark.parametrize("tabstyle,tablength", [("spline", 1024), ("linear", 3000), ("bitmap", 11)])
@pytest.mark.parametrize("entropy", [False, True])
@pytest.mark.parametrize("as_shipped", [0, 1])
def test_pair_bethe_density_parity(fresh_ctx, pkg, orc, tabstyle, tablength, entropy, as_shipped):
    ctx = fresh_ctx
    deck = util.make_deck(tabstyle, tablength, entropy=entropy, **DENS)
    beads = pkg.synth.make_beads(8, seed=41)
    op = util.oracle_pair("table_ucg_bethe_density", deck)
    op.set_compat(as_shipped)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    ctx.set_option("density_proximity_as_shipped", as_shipped)
    util.upload_from_oracle(ctx, sim, beads)
    gp = util.gpu_pair(ctx, "table_ucg_bethe_density", deck)
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    assert not np.isnan(G["f"]).any()
    assert util.bits_equal(G["f"], O["f"])
    assert util.bits_equal(G["scores"], O["scores"]) and util.bits_equal(G["ucgp"], O["ucgp"])
    ev = sim.ev()
    assert abs(eng - ev["eng_vdwl"]) <= 1e-12 * abs(ev["eng_vdwl"])
    assert np.allclose(vir, ev["virial"], rtol=1e-10, atol=1e-8)
    # momentum is conserved (every CV back-force has its reaction, also across periodic images)
    assert np.abs(G["f"].sum(axis=0)).max() < 1e-8 * np.abs(G["f"]).max() * beads.n ** 0.5
    # pass 3 evaluating its tanh itself instead of reading pass 1's (option density_tcache 0): the same bits
    ctx.set_option("density_tcache", 0)
    gp.compute(0, 0)
    assert util.bits_equal(ctx.atoms_download()["f"], O["f"])
    ctx.set_option("density_tcache", 1)
    # the reference's loop shape (sequential sweep, scatter to j): same numbers, other association
    sim0 = util.oracle_sim(beads, op, mode=0)
    sim0.rebuild()
    assert sim0.compute_forces(1, 1) == 0
    R = sim0.arrays()
    assert np.abs(G["f"] - R["f"]).max() <= 1e-10 * np.abs(R["f"]).max()


@pytest.mark.parametrize("slots", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
def test_gather_slots_define_the_canonical_order(fresh_ctx, pkg, orc, slots, style):
    """lanes per bead of the full-row gather kernel (option pair_vrow 0) = interleaved partial sums + fixed tree in the
    oracle's ordered sums"""
    ctx = fresh_ctx
    ctx.set_option("pair_vrow", 0)
    ctx.set_option("gather_slots", slots)
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(9, seed=slots)
    beads.ucgp = np.clip(np.random.default_rng(3).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    op = util.oracle_pair(style, deck, slots=slots)
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    util.upload_from_oracle(ctx, sim, beads)
    gp = util.gpu_pair(ctx, style, deck)
    assert gp.gather_slots == slots
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    for k in ("f", "ucgforce", "scores"):
        assert util.bits_equal(G[k], O[k]), k
    assert abs(eng - sim.ev()["eng_vdwl"]) <= 1e-12 * abs(eng)
    # ... and with the rows read by non-temporal loads (the tuned kernels of one and two lanes per bead have such variants)
    ctx.set_option("stream_rows", 1)
    gp.compute(0, 0)
    gp.check_errors()
    S = ctx.atoms_download()
    ctx.set_option("stream_rows", -1)
    assert util.bits_equal(S["f"], O["f"]) and util.bits_equal(S["scores"], O["scores"])


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
@pytest.mark.parametrize("slots", [0, 1, 4, 16])
def test_interior_and_boundary_launches_add_up(fresh_ctx, pkg, style, slots):
    """ucg_pair_compute_part 1 (workgroups without ghost neighbours) + 2 (the rest) == ucg_pair_compute; slots 0 = the
    virtual-row kernels (option pair_vrow: 512 beads per workgroup), else the full-row gather kernels with that many
    lanes per bead"""
    ctx = fresh_ctx
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(30, seed=4)
    beads.ucgp = np.clip(np.random.default_rng(1).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.set_option("pair_vrow", 0 if slots else 1)
    ctx.set_option("gather_slots", slots if slots else 1)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    ctx.neigh_rebuild()
    gp = util.gpu_pair(ctx, style, deck)
    gp.compute(0, 0)
    full = ctx.atoms_download()
    ctx.force_clear()
    gp.compute_part(1)
    part1 = ctx.atoms_download()
    interior = np.any(part1["f"] != 0.0, axis=1)
    assert interior.sum() < beads.n
    assert gp.sum_fixed == (slots == 0)
    if slots >= 4 or slots == 0:
        assert interior.sum() > 0  # some workgroups are interior (at 1024 beads per workgroup none is, at this size)
    gp.compute_part(2)
    both = ctx.atoms_download()
    for k in ("f", "scores", "ucgforce"):
        assert util.bits_equal(both[k], full[k]), k
    # interior beads were final after part 1
    assert util.bits_equal(part1["f"][interior], full["f"][interior])


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
def test_fma_contracted_kernels_within_tolerance(fresh_ctx, pkg, orc, style):
    """option "fma_contract": the gather kernels compiled with FMA contraction.  NOT bit-identical to the scalar
    reference (each fused multiply-add rounds once instead of twice); stated tolerance: 1e-12 of the largest
    component, per quantity."""
    ctx = fresh_ctx
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(12, seed=19)
    beads.ucgp = np.clip(np.random.default_rng(5).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    op = util.oracle_pair(style, deck)
    op.set_sum_fixed(False)  # the contracted kernels are full-row gather kernels: ordered sums
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    assert sim.compute_forces(0, 0) == 0
    O = sim.arrays()
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.set_option("fma_contract", 1)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    ctx.neigh_rebuild()
    gp = util.gpu_pair(ctx, style, deck)
    gp.compute(0, 0)
    gp.check_errors()
    G = ctx.atoms_download()
    differs = False
    for k in ("f", "scores") + (("ucgforce",) if style == "table_ucgld" else ()):
        scale = np.max(np.abs(O[k]))
        assert np.max(np.abs(G[k] - O[k])) <= 1e-12 * scale, k
        differs |= not util.bits_equal(G[k], O[k])
    assert differs  # it really is a different rounding, not the exact kernels


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 4, 10])
def test_rng_window_follows_the_sequential_stream_when_the_bead_count_changes(fresh_ctx, pkg, batch):
    """fix ucgld/langevin takes one RanMars draw per owned bead per step from ONE sequential stream
    (UCG/fix_ucgld_langevin.cpp:280).  The library generates several steps' worth per launch (option rng_batch); when
    the bead count changes before a window is used up (migration), the stream must continue exactly where the
    consumed draws ended."""
    ctx = fresh_ctx
    dt, seed = 0.002, 48279
    ctx.set_units(1.0, 1.0, 1.0, dt)
    ctx.set_option("rng_batch", batch)
    ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, seed)
    pos = 0
    g2 = np.sqrt(10.0) * np.sqrt(24.0 / 1.0 / dt)  # gfactor2 of init(): sqrt(m_lambda)/ftm2v * sqrt(24 kB/(t_period dt mvv2e))
    for ncell, ncalls in ((6, 3), (7, 12), (5, 2), (7, 1)):
        beads = pkg.synth.make_beads(ncell, seed=3 + ncell)   # ucgvl = 0: the friction term vanishes
        ctx.upload_beads(beads)
        ctx.fix_ucgld_langevin_init(2, beads.ucgml[:3])
        for _ in range(ncalls):
            ctx.force_clear()
            ctx.fix_ucgld_langevin_post_force(0, 0, 10)
            got = ctx.atoms_download()["ucgforce"]
            u = ctx.ranmars_fill(seed, pos, beads.n)
            want = g2 * 1.0 * (u - 0.5)
            assert np.allclose(got, want, rtol=1e-13, atol=0.0), (ncell, pos)
            pos += beads.n


@pytest.mark.gpu
@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
def test_bitmap_file_section_on_the_device(fresh_ctx, pkg, orc, style):
    """a BITMAP section of the table file (2^10 entries in bit order, used verbatim: "match") through the device kernels"""
    ctx = fresh_ctx
    deck = util.make_deck("bitmap", 10, n_file=1024, rmode="BITMAP")
    beads = pkg.synth.make_beads(9, seed=21)
    beads.ucgp = np.clip(np.random.default_rng(4).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    op = util.oracle_pair(style, deck)
    assert op.table_info(0)["match"] == 1
    sim = util.oracle_sim(beads, op, mode=1)
    sim.rebuild()
    assert sim.compute_forces(1, 1) == 0
    O = sim.arrays()
    util.upload_from_oracle(ctx, sim, beads)
    gp = util.gpu_pair(ctx, style, deck)
    eng, vir = gp.compute(1, 1)
    gp.check_errors()
    G = ctx.atoms_download()
    for k in ("f", "scores", "ucgforce"):
        assert util.bits_equal(G[k], O[k]), k
    assert abs(eng - sim.ev()["eng_vdwl"]) <= 1e-12 * abs(eng)
