"""worker for the world_size-2 tests (launched by test_multi_rank.py through subprocess)

  python mp_worker.py cpu <port> <outdir>   -- gloo on CPU: decomposition + transport protocol
  python mp_worker.py gpu <port> <outdir>   -- gloo (host-staged) with both ranks on ONE GPU
  python mp_worker.py gpu_density ...       -- the same with table_ucg_bethe_density (two mid-compute halos)
  python mp_worker.py gpu_lang ...          -- thermostatted run with state switching (per-rank RanMars streams)
  python mp_worker.py gpu_config5 ...       -- BASELINE.json config 5 in small: table_ucg_bethe_density + ucgstate mc +
                                               fix cluster_switch (two actual atom types), decomposed
"""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: E402
import util  # noqa: E402


def main():
    mode, port, outdir = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = port
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = conftest.load_package()
    multi, synth = pkg.multi, pkg.synth
    beads = synth.make_beads(10, seed=5)
    grid = multi.choose_procgrid(world)
    result = {}
    if mode == "cpu":
        tr = multi.Transport(dist, torch.device("cpu"), staged=True)
        sl = slice(rank * beads.n // world, (rank + 1) * beads.n // world)
        x, tag = beads.x[sl], beads.tag[sl]
        dest = multi.owner_rank(x, beads.boxlo, beads.boxhi, grid)
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=world).astype(np.int64)
        rec = np.zeros((len(x), 4))
        rec[:, :3], rec[:, 3] = x[order], tag[order]
        rc = tr.alltoall_counts(counts)
        sb = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())
        rb = tr.alltoall_bytes(sb, counts, rc, 32)
        got = rb.numpy()[: int(rc.sum()) * 32].view(np.float64).reshape(-1, 4)
        lo, hi = multi.sub_box(beads.boxlo, beads.boxhi, grid, rank)
        result = dict(n=len(got), inside=bool(np.all((got[:, :3] >= lo) & (got[:, :3] < hi))), tags=got[:, 3].astype(np.int64),
                      counts=counts, rc=rc, maxflag=tr.allreduce_max(rank), total=tr.allreduce_sum([len(got)])[0])
    elif mode == "gpu_lang":
        # thermostatted run: per-rank RanMars streams (seed + me) with draw windows that have to follow the bead
        # count through migrations
        capi = pkg.capi
        deck = util.make_deck("spline", 1024)
        ctx = capi.Context(0, dt=0.004)
        sl = slice(rank * beads.n // world, (rank + 1) * beads.n // world)
        n = sl.stop - sl.start
        ctx.atoms_upload(n, 0, beads.ntypes, beads.x[sl], beads.v[sl], beads.type[sl], beads.tag[sl], beads.mask[sl],
                         beads.ucgstate[sl], beads.ucgl[sl], beads.ucgvl[sl], beads.ucgml[sl], beads.ucgp[sl], beads.mass)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=0, check=1)
        ctx.set_option("pair_vrow", int(os.environ.get("UCG_TEST_PAIR_VROW", "0")))  # 1: the virtual-row kernels
        pair = util.gpu_pair(ctx, "table_ucgld", deck)
        ctx.set_option("rng_batch", int(os.environ.get("UCG_TEST_RNG_BATCH", "10")))
        ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279, me=rank)
        ctx.fix_ucgstate("mc", 9127, 0.3, me=rank)
        ctx.fix_nve_ucgld_wall_hard(False, 0.1)
        tr = multi.Transport(dist, torch.device("cuda", 0), staged=True)
        sim = multi.RankSim(ctx, pair, tr, grid, use_langevin=True, use_ucgstate=True, integrator="wall")
        sim.setup(360)  # long enough for beads to cross the brick faces (0.54 sigma from the nearest lattice plane)
        counts = [ctx.counts()[0]]
        for _ in range(6):
            sim.run(60)
            counts.append(ctx.counts()[0])
        pair.check_errors()
        A = ctx.atoms_download()
        nl = A["nlocal"]
        result = dict(tag=A["tag"][:nl], x=A["x"][:nl], l=A["ucgl"][:nl], v=A["v"], st=A["ucgstate"][:nl], counts=counts,
                      nrebuild=sim.nrebuild)
        pair.close()
        ctx.close()
    elif mode == "gpu_rccl":
        # one GPU per rank, the library's RCCL transport between DISTINCT devices (tests/test_multi_rank.py skips this
        # where the box has fewer GPUs than ranks)
        capi = pkg.capi
        beads = synth.make_beads(12, seed=5)
        deck = util.make_deck("spline", 1024)
        ctx = capi.Context(rank, dt=0.004)
        sl = slice(rank * beads.n // world, (rank + 1) * beads.n // world)
        n = sl.stop - sl.start
        ctx.atoms_upload(n, 0, beads.ntypes, beads.x[sl], beads.v[sl], beads.type[sl], beads.tag[sl], beads.mask[sl],
                         beads.ucgstate[sl], beads.ucgl[sl], beads.ucgvl[sl], beads.ucgml[sl], beads.ucgp[sl], beads.mass)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=0, check=1)
        pair = util.gpu_pair(ctx, "table_ucgld", deck)
        box = [capi.Context.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        tr = multi.Transport(dist, torch.device("cuda", rank), staged=True)
        sim = multi.RankSim(ctx, pair, tr, grid, use_langevin=False, use_ucgstate=False, rccl_id=box[0])
        assert sim.transport_note is None, sim.transport_note
        sim.setup(40)
        A0 = ctx.atoms_download()
        sim.run(40)
        pair.check_errors()
        A1 = ctx.atoms_download()
        result = dict(transport=ctx.comm_transport(), tag0=A0["tag"], f0=A0["f"], uf0=A0["ucgforce"], s0=A0["scores"],
                      tag1=A1["tag"], x1=A1["x"], l1=A1["ucgl"], nrebuild=sim.nrebuild)
        pair.close()
        ctx.close()
    elif mode == "gpu_fault":
        # a rank-local failure in the step loop (injected on rank 1 at step 7): BOTH ranks must come back from ucg_md_run
        # with an error -- the failing one with its own message, the other told that a peer failed -- instead of one of
        # them blocking in a receive
        capi = pkg.capi
        deck = util.make_deck("spline", 1024)
        ctx = capi.Context(0, dt=0.004)
        sl = slice(rank * beads.n // world, (rank + 1) * beads.n // world)
        n = sl.stop - sl.start
        ctx.atoms_upload(n, 0, beads.ntypes, beads.x[sl], beads.v[sl], beads.type[sl], beads.tag[sl], beads.mask[sl],
                         beads.ucgstate[sl], beads.ucgl[sl], beads.ucgvl[sl], beads.ucgml[sl], beads.ucgp[sl], beads.mass)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=0, check=1)
        ctx.set_option("pair_vrow", int(os.environ.get("UCG_TEST_PAIR_VROW", "0")))  # 1: the virtual-row kernels
        pair = util.gpu_pair(ctx, "table_ucgld", deck)
        tr = multi.Transport(dist, torch.device("cuda", 0), staged=True)
        sim = multi.RankSim(ctx, pair, tr, grid, use_langevin=False, use_ucgstate=False)
        where = os.environ.get("UCG_TEST_FAULT", "step")
        code, msg = 0, ""
        try:
            if where == "setup":  # inside ucg_md_setup, after the re-neighbouring's own agreement
                if rank == 1:
                    ctx.set_option("fault_inject_setup", 1)
                sim.setup(40)
            else:
                sim.setup(40)
                if rank == 1:
                    ctx.set_option("fault_inject_step", 7)
            sim.run(40)
        except capi.UcgError as e:
            code, msg = e.code, e.msg
        result = dict(code=code, msg=msg, ntimestep=ctx.md_info()["ntimestep"])
        pair.close()
        ctx.close()
    elif mode == "gpu_cluster":
        # fix cluster_switch on a decomposed run: labels must equal the single-rank ones; then a short run
        capi = pkg.capi
        deck = util.make_multi_deck(2, "spline", 256)
        mb = util.multi_type_beads(pkg, 10, 2, seed=5, molecule_size=2)
        rates, contacts = pkg.synth.write_cluster_switch_files(deck.workdir, 0.35, [1], [2], [(1, 1)])
        mol_seed = int(mb.molecule[np.flatnonzero(mb.type == 1)[0]])
        ctx = capi.Context(0, dt=0.004)
        sl = slice(rank * mb.n // world, (rank + 1) * mb.n // world)
        n = sl.stop - sl.start
        ctx.atoms_upload(n, 0, mb.ntypes, mb.x[sl], mb.v[sl], mb.type[sl], mb.tag[sl], mb.mask[sl], mb.ucgstate[sl],
                         mb.ucgl[sl], mb.ucgvl[sl], mb.ucgml[sl], mb.ucgp[sl], mb.mass)
        ctx.upload_molecule(mb.molecule[sl])
        ctx.domain_set(mb.boxlo, mb.boxhi, 2.5, 0.3, every=5, delay=0, check=1)
        pair = util.gpu_pair_multi(ctx, "table_ucgld", deck)
        tr = multi.Transport(dist, torch.device("cuda", 0), staged=True)
        sim = multi.RankSim(ctx, pair, tr, grid, use_langevin=False, use_ucgstate=False)
        sim.cluster_switch(mol_seed, 0, 1.15, 4711, 5, rates, contacts)
        sim.setup(30)
        # check_cluster alone (the phases of RankSim._cluster_step up to finalize)
        changed = ctx.cs_sweep(1)
        rounds = 0
        while True:
            ctx.cs_set_array(5, tr.allreduce_array(ctx.cs_array(5), "min"))
            rounds += 1
            if not tr.allreduce_max(changed):
                break
            changed = ctx.cs_sweep(0)
        ctx.cs_finalize()
        labels = ctx.cs_array(0)
        state0, restrict0 = ctx.cs_array(1), ctx.cs_array(2)
        sim.run(30)
        pair.check_errors()
        A = ctx.atoms_download()
        result = dict(labels=labels, state0=state0, restrict0=restrict0, rounds=rounds, tag=A["tag"], type=A["type"],
                      mol=ctx.download_molecule(), vec=ctx.fix_cluster_switch_vector(), state1=ctx.cs_array(1),
                      mol_seed=mol_seed, nrebuild=sim.nrebuild)
        pair.close()
        ctx.close()
    elif mode == "gpu_config5":
        # BASELINE.json config 5 in small: table_ucg_bethe_density (two actual types, both density types) + fix ucgstate mc
        # + fix cluster_switch, decomposed: the density style's two mid-compute halos AND the cluster reductions
        capi = pkg.capi
        deck = util.make_multi_deck(2, "spline", 1024, density=(11.3, 1.5), extra11=0.05)  # ten tables: through L2 + the LDS hot block
        mb = util.multi_type_beads(pkg, 10, 2, seed=5, molecule_size=2)
        rates, contacts = pkg.synth.write_cluster_switch_files(deck.workdir, 0.35, [1], [2], [(1, 1)])
        mol_seed = int(mb.molecule[np.flatnonzero(mb.type == 1)[0]])
        ctx = capi.Context(0, dt=0.002)
        sl = slice(rank * mb.n // world, (rank + 1) * mb.n // world)
        n = sl.stop - sl.start
        ctx.atoms_upload(n, 0, mb.ntypes, mb.x[sl], mb.v[sl], mb.type[sl], mb.tag[sl], mb.mask[sl], mb.ucgstate[sl],
                         mb.ucgl[sl], mb.ucgvl[sl], mb.ucgml[sl], mb.ucgp[sl], mb.mass)
        ctx.upload_molecule(mb.molecule[sl])
        ctx.domain_set(mb.boxlo, mb.boxhi, 2.5, 0.3, every=5, delay=0, check=1)
        pair = util.gpu_pair_multi(ctx, "table_ucg_bethe_density", deck)
        ctx.fix_ucgstate("mc", 9127, 0.3, me=rank)
        tr = multi.Transport(dist, torch.device("cuda", 0), staged=True)
        sim = multi.RankSim(ctx, pair, tr, grid, use_langevin=False, use_ucgstate=True)
        sim.cluster_switch(mol_seed, 0, 1.15, 4711, 5, rates, contacts)
        e0, v0 = sim.setup(30, ntypes=mb.ntypes)
        A0 = ctx.atoms_download()
        labels = None
        if os.environ.get("UCG_TEST_CS_LABELS_AT_SETUP", "1") == "1":
            # check_cluster by hand on the setup's configuration (it also sets the seed cluster's restrict / state flags, as
            # every check_cluster does; the comparison of whole trajectories with orc_world runs without it)
            changed = ctx.cs_sweep(1)
            while True:
                ctx.cs_set_array(5, tr.allreduce_array(ctx.cs_array(5), "min"))
                if not tr.allreduce_max(changed):
                    break
                changed = ctx.cs_sweep(0)
            ctx.cs_finalize()
            labels = ctx.cs_array(0)
        sim.run(30)
        pair.check_errors()
        A = ctx.atoms_download()
        result = dict(tag0=A0["tag"], f0=A0["f"], p0=A0["ucgp"], st0=A0["ucgstate"], e0=e0, cs_state=ctx.cs_array(1),
                      labels=labels, tag=A["tag"], type=A["type"], st=A["ucgstate"], x=A["x"], mol=ctx.download_molecule(),
                      vec=ctx.fix_cluster_switch_vector(), mol_seed=mol_seed, nrebuild=sim.nrebuild, nghost=A["nghost"])
        pair.close()
        ctx.close()
    else:
        capi = pkg.capi
        dt = 0.004
        density = mode == "gpu_density"
        style = "table_ucg_bethe_density" if density else "table_ucgld"
        deck = util.make_deck("spline", 1024, **(dict(density=(11.3, 1.5), extra11=0.05) if density else {}))
        if density:
            dt = 0.002
        ctx = capi.Context(0, dt=dt)
        sl = slice(rank * beads.n // world, (rank + 1) * beads.n // world)
        n = sl.stop - sl.start
        ctx.atoms_upload(n, 0, beads.ntypes, beads.x[sl], beads.v[sl], beads.type[sl], beads.tag[sl], beads.mask[sl],
                         beads.ucgstate[sl], beads.ucgl[sl], beads.ucgvl[sl], beads.ucgml[sl], beads.ucgp[sl], beads.mass)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=0, check=1)
        ctx.set_option("pair_vrow", int(os.environ.get("UCG_TEST_PAIR_VROW", "0")))  # 1: the virtual-row kernels
        pair = util.gpu_pair(ctx, style, deck)
        tr = multi.Transport(dist, torch.device("cuda", 0), staged=True)
        sim = multi.RankSim(ctx, pair, tr, grid, use_langevin=False, use_ucgstate=False)
        e0, v0 = sim.setup(40)
        A0 = ctx.atoms_download()
        lo, hi = multi.sub_box(beads.boxlo, beads.boxhi, grid, rank)
        inside = bool(np.all((A0["x"] >= lo) & (A0["x"] < hi)))
        last = sim.run(40, thermo_every=40)
        pair.check_errors()
        A1 = ctx.atoms_download()
        etot = [e0, last[0]]  # setup() / run() return the totals over the ranks
        th = sim.thermo(last, mass=beads.mass)
        result = dict(thermo=th, tag0=A0["tag"], f0=A0["f"], uf0=A0["ucgforce"], s0=A0["scores"], p0=A0["ucgp"], inside=inside, tag1=A1["tag"],
                      x1=A1["x"], l1=A1["ucgl"], e0=etot[0], e1=etot[1], nrebuild=sim.nrebuild, nghost=A1["nghost"])
        pair.close()
        ctx.close()
    with open(os.path.join(outdir, f"rank{rank}.pkl"), "wb") as fh:
        pickle.dump(result, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
