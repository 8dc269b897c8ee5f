"""N > 1 path: world_size-2 runs.  CPU tier: gloo, decomposition + transport protocol.
GPU tier: both ranks share the one GPU of the box (gloo, host-staged buffers) and the
decomposed run is compared with the single-rank GPU run of the same beads."""
import os
import pickle
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import util

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _launch(mode, world=2, timeout=300):
    out = tempfile.mkdtemp(prefix="ucgmp_")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker.py"), mode, port, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return [pickle.load(open(os.path.join(out, f"rank{r}.pkl"), "rb")) for r in range(world)]


def test_decomposition_helpers(pkg):
    m = pkg.multi
    assert m.choose_procgrid(1) == [1, 1, 1] and m.choose_procgrid(2) == [2, 1, 1]
    assert m.choose_procgrid(4) == [2, 2, 1] and m.choose_procgrid(8) == [2, 2, 2]
    lo, hi = np.zeros(3), np.array([10.0, 12.0, 14.0])
    vol = 0.0
    for me in range(8):
        a, b = m.sub_box(lo, hi, [2, 2, 2], me)
        vol += np.prod(b - a)
        c = 0.5 * (a + b)
        assert m.owner_rank(c[None, :], lo, hi, [2, 2, 2])[0] == me
    assert abs(vol - np.prod(hi)) < 1e-9
    # boundary points belong to the upper brick, the box top to the last one
    assert m.owner_rank(np.array([[5.0, 0.0, 0.0]]), lo, hi, [2, 1, 1])[0] == 1
    assert m.owner_rank(np.array([[4.999999, 0.0, 0.0]]), lo, hi, [2, 1, 1])[0] == 0


def test_world2_gloo_exchange_protocol():
    res = _launch("cpu")
    tags = np.concatenate([r["tags"] for r in res])
    assert sorted(tags.tolist()) == list(range(1, 1001))        # every bead exactly once
    assert all(r["inside"] for r in res)                         # and inside its owner's brick
    assert res[0]["rc"][1] == res[1]["counts"][0] and res[1]["rc"][0] == res[0]["counts"][1]
    assert all(r["maxflag"] == 1 for r in res) and all(r["total"] == 1000 for r in res)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_world2_decomposed_run_matches_single_rank(fresh_ctx, pkg, world):
    """2 x 1 x 1 and 2 x 2 x 1 bricks (four ranks share the one GPU of the test box)"""
    res = _launch("gpu", world=world)
    assert all(r["inside"] for r in res)
    assert all(r["nrebuild"] >= 2 and r["nghost"] > 0 for r in res)
    # single-rank GPU run of the same beads, same settings
    ctx = fresh_ctx
    beads = pkg.synth.make_beads(10, seed=5)
    deck = util.make_deck("spline", 1024)
    ctx.set_units(1.0, 1.0, 1.0, 0.004)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=0, check=1)
    gp = util.gpu_pair(ctx, "table_ucgld", deck)
    ctx.md_attach(gp, nve=True, langevin=False, ucgstate=False)
    ctx.md_setup(40)
    S0 = ctx.atoms_download()
    e_single0 = ctx.md_thermo()["eng_vdwl"]
    ctx.md_run(40, 40)
    S1 = ctx.atoms_download()
    e_single1 = ctx.md_thermo()["eng_vdwl"]

    def by_tag(tag, arr):
        out = np.zeros((beads.n,) + arr.shape[1:])
        out[tag - 1] = arr
        return out

    tag0 = np.concatenate([r["tag0"] for r in res])
    assert sorted(tag0.tolist()) == list(range(1, beads.n + 1))
    for key, skey in (("f0", "f"), ("uf0", "ucgforce"), ("s0", "scores")):
        multi = by_tag(tag0, np.concatenate([r[key] for r in res]))
        single = by_tag(S0["tag"], S0[skey])
        assert np.max(np.abs(multi - single)) <= 1e-11 * np.max(np.abs(single)), key
    assert abs(res[0]["e0"] - e_single0) <= 1e-11 * abs(e_single0)
    tag1 = np.concatenate([r["tag1"] for r in res])
    x_multi = by_tag(tag1, np.concatenate([r["x1"] for r in res]))
    x_single = by_tag(S1["tag"], S1["x"])
    d = x_multi - x_single
    d -= np.round(d / beads.boxhi) * beads.boxhi
    assert np.max(np.abs(d)) < 1e-9
    l_multi = by_tag(tag1, np.concatenate([r["l1"] for r in res]))
    assert np.max(np.abs(l_multi - by_tag(S1["tag"], S1["ucgl"]))) < 1e-9
    assert abs(res[0]["e1"] - e_single1) <= 1e-9 * abs(e_single1)
    # the reduced thermo line is the same on every rank and equals the single-rank totals
    # ... and against the ORACLE's single-rank trajectory of the same beads (reference-order loop), by tag
    op = util.oracle_pair("table_ucgld", deck)
    osim = util.oracle_sim(beads, op, mode=0, dt=0.004, nve=True, every=2)
    assert osim.setup(40) == 0
    O0 = osim.arrays()
    for key, okey in (("f0", "f"), ("uf0", "ucgforce"), ("s0", "scores")):
        multi = by_tag(tag0, np.concatenate([r[key] for r in res]))
        ref = by_tag(O0["tag"], O0[okey])
        assert np.max(np.abs(multi - ref)) <= 1e-11 * np.max(np.abs(ref)), ("oracle", key)
    assert osim.run(40, 40) == 0
    O1 = osim.arrays()
    d = x_multi - by_tag(O1["tag"], O1["x"])
    d -= np.round(d / beads.boxhi) * beads.boxhi
    assert np.max(np.abs(d)) < 1e-9
    assert np.max(np.abs(l_multi - by_tag(O1["tag"], O1["ucgl"]))) < 1e-9
    assert abs(res[0]["e1"] - osim.ev()["eng_vdwl"]) <= 1e-9 * abs(e_single1)
    th = res[0]["thermo"]
    for r in res[1:]:
        assert r["thermo"]["eng_vdwl"] == th["eng_vdwl"] and r["thermo"]["natoms"] == th["natoms"] == beads.n
    assert abs(th["eng_vdwl"] - e_single1) <= 1e-9 * abs(e_single1)
    n = S1["nlocal"]
    ke_single = 0.5 * float(np.sum(beads.mass[S1["type"][:n]] * np.sum(S1["v"][:n] ** 2, axis=1)))
    assert abs(th["ke"] - ke_single) <= 1e-8 * ke_single
    assert th["state1"] == float(S1["ucgstate"][:n].sum())
    assert abs(th["sum_lambda"] - float(S1["ucgl"][:n].sum())) <= 1e-8 * beads.n
    assert np.allclose(th["virial"], ctx.md_thermo()["virial"], rtol=1e-8, atol=1e-6)


@pytest.mark.gpu
def test_world2_density_style_matches_single_rank(fresh_ctx, pkg):
    """table_ucg_bethe_density decomposed: the priors and the CV forces of ghosts cross ranks between the passes"""
    res = _launch("gpu_density")
    assert all(r["inside"] for r in res)
    assert all(r["nrebuild"] >= 2 and r["nghost"] > 0 for r in res)
    ctx = fresh_ctx
    beads = pkg.synth.make_beads(10, seed=5)
    deck = util.make_deck("spline", 1024, density=(11.3, 1.5), extra11=0.05)
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=0, check=1)
    gp = util.gpu_pair(ctx, "table_ucg_bethe_density", deck)
    ctx.md_attach(gp, nve=True, langevin=False, ucgstate=False)
    ctx.md_setup(40)
    S0 = ctx.atoms_download()
    e_single0 = ctx.md_thermo()["eng_vdwl"]
    ctx.md_run(40, 40)
    S1 = ctx.atoms_download()

    def by_tag(tag, arr):
        out = np.zeros((beads.n,) + arr.shape[1:])
        out[tag - 1] = arr
        return out

    tag0 = np.concatenate([r["tag0"] for r in res])
    assert sorted(tag0.tolist()) == list(range(1, beads.n + 1))
    for key, skey in (("f0", "f"), ("p0", "ucgp")):
        multi = by_tag(tag0, np.concatenate([r[key] for r in res]))
        single = by_tag(S0["tag"], S0[skey])
        assert np.max(np.abs(multi - single)) <= 1e-10 * np.max(np.abs(single)), key
    assert abs(res[0]["e0"] - e_single0) <= 1e-10 * abs(e_single0)
    tag1 = np.concatenate([r["tag1"] for r in res])
    d = by_tag(tag1, np.concatenate([r["x1"] for r in res])) - by_tag(S1["tag"], S1["x"])
    d -= np.round(d / beads.boxhi) * beads.boxhi
    assert np.max(np.abs(d)) < 1e-8


@pytest.mark.gpu
def test_world2_cluster_switch(fresh_ctx, pkg):
    """fix cluster_switch decomposed: the cluster labels do not depend on the decomposition; the switching keeps
    molecules whole and its bookkeeping consistent (which rank draws for a molecule follows the reference: the rank
    holding the majority of its atoms, each with its own RanPark stream, so the draws themselves differ)"""
    res = _launch("gpu_cluster")
    deck = util.make_multi_deck(2, "spline", 256)
    mb = util.multi_type_beads(pkg, 10, 2, seed=5, molecule_size=2)
    rates, contacts = pkg.synth.write_cluster_switch_files(deck.workdir, 0.35, [1], [2], [(1, 1)])
    mol_seed = res[0]["mol_seed"]
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.004)
    ctx.upload_beads(mb)
    ctx.domain_set(mb.boxlo, mb.boxhi, 2.5, 0.3, every=5, delay=0, check=1)
    ctx.neigh_rebuild()
    ctx.fix_cluster_switch(mol_seed, 0, 1.15, 4711, 5, rates, contacts)
    ctx.fix_cluster_switch_check_cluster()
    S = ctx.fix_cluster_switch_arrays()
    for r in res:
        assert np.array_equal(r["labels"], S["mol_cluster"])
        assert np.array_equal(r["state0"], S["mol_state"]) and np.array_equal(r["restrict0"], S["mol_restrict"])
        assert r["rounds"] >= 2  # at least one reduction round was needed
    assert np.array_equal(res[0]["labels"], res[1]["labels"])
    # after 30 steps with switching every 5: every bead exactly once, molecules whole, bookkeeping consistent
    tag = np.concatenate([r["tag"] for r in res])
    typ = np.concatenate([r["type"] for r in res])
    mol = np.concatenate([r["mol"] for r in res])
    assert sorted(tag.tolist()) == list(range(1, mb.n + 1))
    assert np.array_equal(mol, mb.molecule[tag - 1])  # molecule ids travelled with the beads
    assert set(np.unique(typ)) <= {1, 2}
    for m in np.unique(mol):
        assert len(np.unique(typ[mol == m])) == 1
    assert np.array_equal(res[0]["vec"], res[1]["vec"]) and np.array_equal(res[0]["state1"], res[1]["state1"])
    att, suc = res[0]["vec"][0], res[0]["vec"][1]
    assert att > 100 and 0 < suc < att
    assert (typ != mb.type[tag - 1]).sum() > 0
    # mol_state of every molecule OUTSIDE the seed's cluster says which type its atoms carry
    st = res[0]["state1"]
    t_of_mol = {int(m): int(typ[mol == m][0]) for m in np.unique(mol)}
    labels_now = res[0]["labels"]
    assert all(r["nrebuild"] >= 6 for r in res)
    checked = 0
    for m, t in t_of_mol.items():
        if st[m] in (0, 1) and labels_now[m] != labels_now[mol_seed]:
            checked += 1
    assert checked > 10


@pytest.mark.gpu
def test_world2_thermostatted_run_is_reproducible(pkg, monkeypatch):
    """fix ucgld/langevin + fix ucgstate mc on two ranks: beads migrate, so the ranks' bead counts -- and with them the
    number of RanMars draws per step -- change during the run.  A run with draw windows of ten steps must agree bit
    for bit with one that launches the generator every step (the sequential stream of the reference)"""
    monkeypatch.setenv("UCG_TEST_RNG_BATCH", "10")
    a = _launch("gpu_lang", world=2)
    monkeypatch.setenv("UCG_TEST_RNG_BATCH", "1")
    b = _launch("gpu_lang", world=2)
    n = sum(len(r["tag"]) for r in a)
    assert n == 1000 and sorted(np.concatenate([r["tag"] for r in a]).tolist()) == list(range(1, n + 1))
    assert any(len(set(r["counts"])) > 1 for r in a), "no migration happened: the test does not exercise the windows"
    assert all(r["nrebuild"] >= 3 for r in a)
    for ra, rb in zip(a, b):
        assert ra["counts"] == rb["counts"] and np.array_equal(ra["tag"], rb["tag"]) and np.array_equal(ra["st"], rb["st"])
        for k in ("x", "l", "v"):
            assert util.bits_equal(ra[k], rb[k]), k
    lam = np.concatenate([r["l"] for r in a])
    assert lam.min() >= 0.0 and lam.max() <= 1.0 and 0 < np.concatenate([r["st"] for r in a]).sum() < n


@pytest.mark.gpu
def test_a_rank_local_failure_stops_every_rank_together(pkg):
    """a failure on one rank between two collectives must not leave its peer blocked in a receive.  The
    failing rank keeps taking part in the halo until the next status agreement (the re-neighbour decision's all-reduce,
    here at the next even step), where both ranks return an error from ucg_md_run"""
    res = _launch("gpu_fault", world=2, timeout=240)
    assert res[1]["code"] == 1 and "injected" in res[1]["msg"]
    assert res[0]["code"] != 0 and "another rank failed" in res[0]["msg"]
    assert res[0]["ntimestep"] == res[1]["ntimestep"] == 8  # injected at step 7, agreed at the decision of step 8


@pytest.mark.gpu
def test_a_rank_local_failure_in_the_setup_stops_every_rank_together(pkg, monkeypatch):
    """the same promise for ucg_md_setup -- a rank that fails there (here: from its start) sends nothing in
    the re-neighbouring, and BOTH ranks return from its closing agreement; neither goes on into a collective alone"""
    monkeypatch.setenv("UCG_TEST_FAULT", "setup")
    res = _launch("gpu_fault", world=2, timeout=240)
    assert res[1]["code"] == 1 and "injected" in res[1]["msg"]
    assert res[0]["code"] != 0 and "another rank" in res[0]["msg"]
    assert res[0]["ntimestep"] == res[1]["ntimestep"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("vrow", [0, 1])
def test_world2_thermostatted_run_equals_the_decomposed_oracle_bit_for_bit(pkg, orc, vrow, monkeypatch):
    """fix ucgld/langevin + fix ucgstate mc + fix nve/ucgld/wall/hard on two ranks against the oracle's statement of the
    same decomposed run (oracle/orc_md.c: orc_world -- bricks, per-rank bead order, RanMars(seed + me) streams drawn in
    local order, migration): every rank's beads in its local order, positions, velocities, lambda and states after 360
    steps with migration, bit for bit; and the bead counts of the ranks along the run.  vrow = 1: the virtual-row kernels
    (option pair_vrow) against the oracle's fixed sums -- which depend on no order, the decomposition included"""
    monkeypatch.setenv("UCG_TEST_PAIR_VROW", str(vrow))
    res = _launch("gpu_lang", world=2)
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(10, seed=5)
    op = util.oracle_pair("table_ucgld", deck)
    op.set_sum_fixed(bool(vrow))
    w = orc.World(beads, [2, 1, 1])
    w.set_run_params(dt=0.004, every=2, delay=0, check=1)
    w.attach(op, langevin=(1.0, 1.0, 1.0, 48279), nve="wall", ucgstate=("mc", 9127, 0.3))
    assert w.setup(360) == 0
    counts = [[w.rank_arrays(r)["nlocal"]] for r in range(2)]
    for _ in range(6):
        assert w.run(60) == 0
        for r in range(2):
            counts[r].append(w.rank_arrays(r)["nlocal"])
    assert any(len(set(c)) > 1 for c in counts)
    for r in range(2):
        O = w.rank_arrays(r)
        G = res[r]
        assert G["counts"] == counts[r]
        assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["st"], O["ucgstate"])
        assert util.bits_equal(G["x"], O["x"]) and util.bits_equal(G["l"], O["ucgl"]) and util.bits_equal(G["v"], O["v"])
        assert G["nrebuild"] == w.rank_info(r)["nrebuild"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,vrow", [(2, 0), (4, 0), (3, 0), (2, 1)])
def test_decomposed_forces_and_trajectory_equal_the_decomposed_oracle_bit_for_bit(pkg, orc, world, vrow, monkeypatch):
    """2 x 1 x 1, 2 x 2 x 1 and 3 x 1 x 1 bricks (three bricks along x: a rank's two x-neighbours are different ranks, as are
    all seven neighbours of a rank of the 2 x 2 x 2 grid of an 8-GPU run; the test box allows six processes on its GPU, this
    process included, so four ranks is the most a test starts): forces, ucgforce and scores of every rank at setup and positions / lambda after 40
    steps with re-neighbouring, in each rank's local order, against orc_world -- the canonical order of a decomposed run
    is a function of the decomposition, and the oracle states it (vrow = 1: the virtual-row kernels, whose fixed sums
    are not)"""
    monkeypatch.setenv("UCG_TEST_PAIR_VROW", str(vrow))
    res = _launch("gpu", world=world)
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(10, seed=5)
    op = util.oracle_pair("table_ucgld", deck)
    op.set_sum_fixed(bool(vrow))
    w = orc.World(beads, pkg.multi.choose_procgrid(world))
    w.set_run_params(dt=0.004, every=2, delay=0, check=1)
    w.attach(op, langevin=None, nve=True, ucgstate=None)
    assert w.setup(40) == 0
    for r in range(world):
        O, G = w.rank_arrays(r), res[r]
        assert np.array_equal(G["tag0"], O["tag"])
        assert util.bits_equal(G["f0"], O["f"]) and util.bits_equal(G["uf0"], O["ucgforce"]) and util.bits_equal(G["s0"], O["scores"])
    assert abs(res[0]["e0"] - w.ev()["eng_vdwl"]) <= 1e-12 * abs(res[0]["e0"])
    assert w.run(40, 40) == 0
    for r in range(world):
        O, G = w.rank_arrays(r), res[r]
        assert np.array_equal(G["tag1"], O["tag"])
        assert util.bits_equal(G["x1"], O["x"]) and util.bits_equal(G["l1"], O["ucgl"])
        assert G["nrebuild"] == w.rank_info(r)["nrebuild"] and G["nghost"] == O["nghost"]


@pytest.mark.gpu
def test_world2_density_style_equals_the_decomposed_oracle_bit_for_bit(pkg, orc):
    """table_ucg_bethe_density on two ranks -- its two mid-compute halos carry the ghosts' priors and CV forces between the
    ranks -- against orc_world's lockstep passes: forces and posteriors of every rank at setup, positions after 40 steps"""
    res = _launch("gpu_density")
    deck = util.make_deck("spline", 1024, density=(11.3, 1.5), extra11=0.05)
    beads = pkg.synth.make_beads(10, seed=5)
    op = util.oracle_pair("table_ucg_bethe_density", deck)
    w = orc.World(beads, [2, 1, 1])
    w.set_run_params(dt=0.002, every=2, delay=0, check=1)
    w.attach(op, langevin=None, nve=True, ucgstate=None)
    assert w.setup(40) == 0
    for r in range(2):
        O, G = w.rank_arrays(r), res[r]
        assert np.array_equal(G["tag0"], O["tag"])
        assert util.bits_equal(G["f0"], O["f"]) and util.bits_equal(G["p0"], O["ucgp"])
    assert abs(res[0]["e0"] - w.ev()["eng_vdwl"]) <= 1e-12 * abs(res[0]["e0"])
    assert w.run(40, 40) == 0
    for r in range(2):
        O, G = w.rank_arrays(r), res[r]
        assert np.array_equal(G["tag1"], O["tag"]) and util.bits_equal(G["x1"], O["x"])
        assert G["nrebuild"] == w.rank_info(r)["nrebuild"]


@pytest.mark.gpu
def test_world2_config5_density_with_cluster_switch_vs_oracle(pkg, orc):
    """BASELINE.json config 5 in small (table_ucg_bethe_density + fix ucgstate mc + fix cluster_switch on two actual atom
    types), two ranks, against the ORACLE's single-rank run of the same beads: forces, posteriors and states at setup by
    tag, the cluster labels (independent of the decomposition), then 30 steps with switching every 5"""
    res = _launch("gpu_config5")
    deck = util.make_multi_deck(2, "spline", 1024, density=(11.3, 1.5), extra11=0.05)  # ten tables: through L2 + the LDS hot block
    mb = util.multi_type_beads(pkg, 10, 2, seed=5, molecule_size=2)
    rates, contacts = pkg.synth.write_cluster_switch_files(deck.workdir, 0.35, [1], [2], [(1, 1)])
    mol_seed = res[0]["mol_seed"]
    op = util.oracle_pair_multi("table_ucg_bethe_density", deck)
    sim = util.oracle_sim(mb, op, mode=1, dt=0.002, nve=True, ucgstate=("mc", 9127, 0.3), every=5)
    sim.cluster_switch(mol_seed, 0, 1.15, 4711, 5, rates, contacts)
    assert sim.setup(30) == 0
    O = sim.arrays()

    def by_tag(tag, arr):
        out = np.zeros((mb.n,) + np.asarray(arr).shape[1:])
        out[np.asarray(tag) - 1] = arr
        return out

    tag0 = np.concatenate([r["tag0"] for r in res])
    assert sorted(tag0.tolist()) == list(range(1, mb.n + 1))
    f_multi, f_orc = by_tag(tag0, np.concatenate([r["f0"] for r in res])), by_tag(O["tag"], O["f"])
    assert np.max(np.abs(f_multi - f_orc)) <= 1e-10 * np.max(np.abs(f_orc))
    p_multi, p_orc = by_tag(tag0, np.concatenate([r["p0"] for r in res])), by_tag(O["tag"], O["ucgp"])
    assert np.max(np.abs(p_multi - p_orc)) <= 1e-10
    assert abs(res[0]["e0"] - sim.ev()["eng_vdwl"]) <= 1e-10 * abs(sim.ev()["eng_vdwl"])
    # cluster labels: the oracle's check_cluster on the same configuration
    L = orc.lib()
    cs = L.orc_sim_cs(sim.h)
    assert L.orc_cs_check_cluster(cs, L.orc_sim_atoms(sim.h), L.orc_sim_molecule(sim.h), L.orc_sim_full_list(sim.h)) == 0
    lab = sim.cs_arrays()["mol_cluster"]
    for r in res:
        assert np.array_equal(r["labels"], lab)
    assert 1 < int((lab == lab[mol_seed]).sum()) < mb.molecule.max()
    # after the run: every bead once, molecules whole (wholly ON or OFF), switching happened, states are 0 / 1
    tag = np.concatenate([r["tag"] for r in res])
    typ = np.concatenate([r["type"] for r in res])
    mol = np.concatenate([r["mol"] for r in res])
    st = np.concatenate([r["st"] for r in res])
    assert sorted(tag.tolist()) == list(range(1, mb.n + 1))
    assert np.array_equal(mol, mb.molecule[tag - 1])
    for m in np.unique(mol):
        assert len(np.unique(typ[mol == m])) == 1
    assert set(np.unique(st)) <= {0, 1} and 0 < st.sum() < mb.n
    assert np.array_equal(res[0]["vec"], res[1]["vec"])
    assert res[0]["vec"][0] > 100 and 0 < res[0]["vec"][1] < res[0]["vec"][0]
    assert (typ != mb.type[tag - 1]).sum() > 0
    assert all(r["nrebuild"] >= 6 and r["nghost"] > 0 for r in res)
    assert np.all(np.isfinite(np.concatenate([r["x"] for r in res])))


@pytest.mark.gpu
def test_world2_config5_with_cluster_switch_equals_the_decomposed_oracle_bit_for_bit(pkg, orc, monkeypatch):
    """the same run against orc_world, which states what a decomposed run of fix cluster_switch does (labels reduced between
    the ranks' sweeps, each rank deciding the molecules it holds from its own RanPark stream in ascending molecule id, the
    decisions reduced, the ghosts taking the new types): forces and posteriors at setup, then after 30 steps with switching
    every 5 -- six rounds of decisions feeding back into the dynamics -- types, states, positions and the fix's statistics of
    every rank, bit for bit"""
    monkeypatch.setenv("UCG_TEST_CS_LABELS_AT_SETUP", "0")
    res = _launch("gpu_config5")
    deck = util.make_multi_deck(2, "spline", 1024, density=(11.3, 1.5), extra11=0.05)
    mb = util.multi_type_beads(pkg, 10, 2, seed=5, molecule_size=2)
    rates, contacts = pkg.synth.write_cluster_switch_files(deck.workdir, 0.35, [1], [2], [(1, 1)])
    op = util.oracle_pair_multi("table_ucg_bethe_density", deck)
    w = orc.World(mb, pkg.multi.choose_procgrid(2))
    w.set_run_params(dt=0.002, every=5, delay=0, check=1)
    w.attach(op, langevin=None, nve=True, ucgstate=("mc", 9127, 0.3))
    w.cluster_switch(res[0]["mol_seed"], 0, 1.15, 4711, 5, rates, contacts)
    assert w.setup(30) == 0
    for r in range(2):
        O, G = w.rank_arrays(r), res[r]
        assert np.array_equal(G["tag0"], O["tag"])
        assert util.bits_equal(G["f0"], O["f"]) and util.bits_equal(G["p0"], O["ucgp"])
    assert w.run(30, 0) == 0
    for r in range(2):
        O, G = w.rank_arrays(r), res[r]
        assert np.array_equal(G["tag"], O["tag"]) and np.array_equal(G["type"], O["type"]) and np.array_equal(G["st"], O["ucgstate"])
        assert util.bits_equal(G["x"], O["x"])
        ca, cst = w.rank_cs(r)
        assert np.array_equal(G["vec"], cst) and np.array_equal(G["cs_state"], ca["mol_state"])
        assert G["nrebuild"] == w.rank_info(r)["nrebuild"]
    assert res[0]["vec"][1] > 0 and (np.concatenate([r["type"] for r in res]) != mb.type[np.concatenate([r["tag"] for r in res]) - 1]).sum() > 0


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_prints_one_json_line():
    """`python bench.py --gpus 2 ...` exactly as the driver calls it (no torchrun, no rendezvous environment): the parent
    starts the ranks before touching the GPU, relays rank 0's single JSON line, and fails when a rank fails.  On this
    one-GPU box the two ranks share the device through the callback communicator (flagged in config.parallelism)."""
    import json

    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--ncell", "14", "--steps", "20",
                        "--warmup", "5", "--equilibrate", "10"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["unit"] == "timesteps/s" and d["value"] > 0
    assert d["scaling"] == "strong" and d["config"]["beads"] == 14 ** 3 and "2x1x1" in d["config"]["parallelism"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    # a wrong option makes a rank fail: the launcher must not hang and must return non-zero
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--ncell", "3", "--steps", "2",
                        "--warmup", "1", "--equilibrate", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0


@pytest.mark.gpu
def test_bench_refuses_to_report_a_host_staged_run_as_the_scaling_result():
    """One GPU per rank but no usable librccl (here: UCG_RCCL_LIBRARY names a file that does not exist): every rank takes
    the callback communicator together -- and bench.py exits NON-ZERO, because a number measured over host-staged gloo must
    not stand in a scaling record as an xGMI one.  UCG_BENCH_ALLOW_HOST_STAGED=1 lets the run complete;
    its JSON line then says "rccl": false, "rccl_nranks": 0.  With RCCL present the same run reports rccl_nranks = 1
    (ncclCommCount of the attached communicator)."""
    import json

    root = os.path.dirname(HERE)
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--ncell", "14", "--steps", "20", "--warmup", "5", "--equilibrate", "10",
           "--no-cpu-baseline"]
    env = dict(base, UCG_FORCE_MULTI="1", UCG_RCCL_LIBRARY="/nonexistent/librccl.so.1", MASTER_PORT="29571")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "RCCL did not attach" in r.stderr, r.stderr[-3000:]
    env["UCG_BENCH_ALLOW_HOST_STAGED"] = "1"
    env["MASTER_PORT"] = "29572"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["value"] > 0 and d["rccl"] is False and d["rccl_nranks"] == 0
    assert "RCCL transport unavailable" in d["config"]["parallelism"] and "host-staged" in d["config"]["parallelism"]
    env = dict(base, UCG_FORCE_MULTI="1", MASTER_PORT="29573")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert d["rccl"] is True and d["rccl_nranks"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_on_separate_gpus_equal_the_decomposed_oracle_bit_for_bit(pkg, orc, world):
    """The first REAL N > 1 run must check bits, not only speed: `world` ranks, one GPU each, the library's
    RCCL transport (ucg_comm_attach_rccl: grouped ncclSend / ncclRecv between distinct devices), against orc_world -- forces,
    ucgforce, scores at setup and positions / lambda after 40 steps with re-neighbouring, every rank in its local order.
    Skipped where the box has fewer GPUs than ranks (RCCL refuses two ranks per device): the driver's 8-GPU node runs it."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (RCCL: one rank per device), this box has {torch.cuda.device_count()}")
    res = _launch("gpu_rccl", world=world, timeout=600)
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(12, seed=5)
    op = util.oracle_pair("table_ucgld", deck)
    w = orc.World(beads, pkg.multi.choose_procgrid(world))
    w.set_run_params(dt=0.004, every=2, delay=0, check=1)
    w.attach(op, langevin=None, nve=True, ucgstate=None)
    assert w.setup(40) == 0
    for r in range(world):
        O, G = w.rank_arrays(r), res[r]
        assert G["transport"]["rccl"] and G["transport"]["rccl_nranks"] == world
        assert np.array_equal(G["tag0"], O["tag"])
        assert util.bits_equal(G["f0"], O["f"]) and util.bits_equal(G["uf0"], O["ucgforce"]) and util.bits_equal(G["s0"], O["scores"])
    assert w.run(40) == 0
    for r in range(world):
        O, G = w.rank_arrays(r), res[r]
        assert np.array_equal(G["tag1"], O["tag"])
        assert util.bits_equal(G["x1"], O["x"]) and util.bits_equal(G["l1"], O["ucgl"])


@pytest.mark.gpu
def test_fixed_sums_do_not_depend_on_the_decomposition_and_ordered_sums_do(fresh_ctx, pkg, monkeypatch):
    """What option pair_vrow is FOR.  A bead's force / ucgforce / scores are sums over its
    neighbours; the default kernels add them in row order -- the canonical order, a function of the bead order and hence of
    the decomposition -- so the same bead gets different last bits on one rank and on two.  With pair_vrow every term is an
    integer image (ucg_pair_sum_fixed) and the sums are order-free: the SAME beads on one rank and on two ranks of a
    2 x 1 x 1 grid have bit-identical sums at setup, compared tag by tag.  (The price: 6 % / 3 % in the kernel)"""
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(10, seed=5)
    out = {}
    for vrow in (1, 0):
        monkeypatch.setenv("UCG_TEST_PAIR_VROW", str(vrow))
        res = _launch("gpu", world=2)
        tag2 = np.concatenate([r["tag0"] for r in res])
        o2 = np.argsort(tag2)
        two = {k: np.concatenate([r[k] for r in res])[o2] for k in ("f0", "uf0", "s0")}
        ctx = pkg.capi.Context(-1, dt=0.004)
        try:
            ctx.set_option("pair_vrow", vrow)
            ctx.upload_beads(beads)
            ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=0, check=1)
            ctx.neigh_rebuild()
            gp = util.gpu_pair(ctx, "table_ucgld", deck)
            assert bool(gp.sum_fixed) == bool(vrow)
            gp.compute(0, 0)
            gp.check_errors()
            A = ctx.atoms_download()
            o1 = np.argsort(A["tag"])
            one = {"f0": A["f"][o1], "uf0": A["ucgforce"][o1], "s0": A["scores"][o1]}
            gp.close()
        finally:
            ctx.close()
        assert np.array_equal(np.sort(tag2), A["tag"][o1])
        out[vrow] = {k: util.bits_equal(one[k], two[k]) for k in one}
        close = max(np.max(np.abs(one[k] - two[k])) / np.max(np.abs(one[k])) for k in one)
        assert close < 1e-11  # the same physics either way
    assert all(out[1].values()), out[1]        # fixed sums: bit for bit the same on 1 and on 2 ranks
    assert not all(out[0].values()), out[0]    # ordered sums: the canonical order follows the decomposition
