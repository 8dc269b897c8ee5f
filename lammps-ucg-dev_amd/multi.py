"""Spatial decomposition across the GPUs of one node: one process per GPU.

Replaces, for the resident path, what upstream LAMMPS' CommBrick does with the field lists of
``UCG/atom_vec_ucg.cpp:66-82``: bead migration (``exchange``), ghost construction
(``borders``) and the per-step forward halo (``forward_comm``: x, ucgstate, ucgl, ucgp).
The gather kernels accumulate nothing on ghosts, so there is NO reverse halo.

Device work (count / pack / unpack, sorting, binning, lists, forces, fixes) is the C ABI's;
this module only sequences it and moves the packed buffers with ``torch.distributed``:
backend ``nccl`` (= RCCL over xGMI) on a multi-GPU node -- one ``all_to_all_single`` per halo,
i.e. direct neighbour exchange on the point-to-point links, not LAMMPS' staged x/y/z
forwarding -- or, for tests on a single GPU shared by the ranks, ``gloo`` with host staging.
"""
from __future__ import annotations

import os

import numpy as np


def choose_procgrid(world: int):
    """2x1x1, 2x2x1, 2x2x2, ... : factor the rank count over x, y, z as evenly as possible"""
    grid = [1, 1, 1]
    n, d, f = world, 0, 2
    factors = []
    while n > 1:
        while n % f == 0:
            factors.append(f)
            n //= f
        f += 1
    for f in sorted(factors, reverse=True):
        i = int(np.argmin(grid))
        grid[i] *= f
    grid.sort(reverse=True)
    return grid


def rank_loc(me: int, grid):
    return me % grid[0], (me // grid[0]) % grid[1], me // (grid[0] * grid[1])


def sub_box(boxlo, boxhi, grid, me):
    """the brick of rank ``me`` -- same expression as the library / LAMMPS (boxlo + prd*i/p)"""
    lo, hi = np.zeros(3), np.zeros(3)
    loc = rank_loc(me, grid)
    for d in range(3):
        prd = boxhi[d] - boxlo[d]
        lo[d] = boxlo[d] + prd * loc[d] / grid[d]
        hi[d] = boxhi[d] if loc[d] + 1 >= grid[d] else boxlo[d] + prd * (loc[d] + 1) / grid[d]
    return lo, hi


def owner_rank(x, boxlo, boxhi, grid):
    """rank owning each (wrapped) position, vectorised"""
    x = np.asarray(x)
    loc = []
    for d in range(3):
        prd = boxhi[d] - boxlo[d]
        l = np.zeros(len(x), dtype=np.int64)
        for i in range(1, grid[d]):
            l[x[:, d] >= boxlo[d] + prd * i / grid[d]] = i
        loc.append(l)
    return loc[0] + grid[0] * (loc[1] + grid[1] * loc[2])


class Transport:
    """all_to_all of packed device buffers + small reductions over torch.distributed"""

    def __init__(self, dist, device, staged: bool):
        import torch

        self.torch = torch
        self.dist = dist
        self.device = device
        self.staged = staged  # gloo: stage through the host
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        # small HOST messages (counts, flags): either a gloo side group on the loopback interface (no device round
        # trip) or the main backend with device tensors.  Which one is faster depends on the rank count, so both are
        # timed once here and every rank takes the same, faster one (UCG_HOST_GROUP=0 / 1 forces the choice).
        self.host_group = None
        self.use_host = False
        self.small_msg_us = None
        want = os.environ.get("UCG_HOST_GROUP", "auto")
        if not staged and want != "0":
            try:
                # one node: the loopback interface is always there (the host name may not resolve)
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                self.host_group = dist.new_group(backend="gloo")
            except Exception:  # noqa: BLE001
                self.host_group = None
            if self.host_group is not None:
                self.use_host = True if want == "1" else self._host_group_is_faster()

    def _host_group_is_faster(self):
        """time the two routes for the two small collectives of a re-neighbouring step; all ranks get one answer"""
        import time

        t = self.torch
        counts = np.zeros(self.world, dtype=np.int64)
        times = []
        for use_host in (True, False):
            self.use_host = use_host
            for _ in range(3):  # warm up connections / communicators
                self.alltoall_counts(counts)
                self.allreduce_max(0)
            self.dist.barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                self.alltoall_counts(counts)
                self.allreduce_max(0)
            times.append((time.perf_counter() - t0) / 10)
        x = t.tensor(times, dtype=t.float64, device=self.device)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
        both = x.cpu().numpy() / self.world
        self.small_msg_us = dict(host_gloo=float(both[0]) * 1e6, device_main=float(both[1]) * 1e6)
        return bool(both[0] <= both[1])

    def alltoall_counts(self, counts):
        t = self.torch
        if self.use_host:
            send = t.from_numpy(np.ascontiguousarray(counts, dtype=np.int64))
            recv = t.empty_like(send)
            self.dist.all_to_all_single(recv, send, group=self.host_group)
            return recv.numpy()
        send = t.tensor(np.asarray(counts, dtype=np.int64), device="cpu" if self.staged else self.device)
        recv = t.empty_like(send)
        self.dist.all_to_all_single(recv, send)
        return recv.cpu().numpy()

    def alltoall_bytes(self, sendbuf, send_counts, recv_counts, rec_bytes):
        """sendbuf: uint8 device tensor grouped by destination; returns the uint8 device tensor received"""
        t = self.torch
        in_splits = [int(c) * rec_bytes for c in send_counts]
        out_splits = [int(c) * rec_bytes for c in recv_counts]
        nout = sum(out_splits)
        if self.staged:
            s = sendbuf[: sum(in_splits)].cpu()
            r = t.empty(nout, dtype=t.uint8)
            self.dist.all_to_all_single(r, s, out_splits, in_splits)
            return r.to(self.device)
        r = t.empty(max(nout, 1), dtype=t.uint8, device=self.device)
        self.dist.all_to_all_single(r[:nout], sendbuf[: sum(in_splits)], out_splits, in_splits)
        return r

    def alltoall_bytes_begin(self, sendbuf, send_counts, recv_counts, rec_bytes):
        """like alltoall_bytes, but returns (handle, tensor) at once on a GPU-native transport: the collective
        runs on the library's own stream while the caller's stream keeps computing; handle.wait() orders the
        caller's stream after it.  Host-staged transports complete before returning (handle None)."""
        if self.staged:
            return None, self.alltoall_bytes(sendbuf, send_counts, recv_counts, rec_bytes)
        t = self.torch
        in_splits = [int(c) * rec_bytes for c in send_counts]
        out_splits = [int(c) * rec_bytes for c in recv_counts]
        nout = sum(out_splits)
        r = t.empty(max(nout, 1), dtype=t.uint8, device=self.device)
        work = self.dist.all_to_all_single(r[:nout], sendbuf[: sum(in_splits)], out_splits, in_splits, async_op=True)
        return work, r

    def allreduce_array(self, arr, op: str):
        """elementwise "max" / "min" / "sum" of an int64 array over the ranks (small per-molecule arrays)"""
        t = self.torch
        x = t.tensor(np.asarray(arr, dtype=np.int64), device="cpu" if self.staged else self.device)
        self.dist.all_reduce(x, op={"max": self.dist.ReduceOp.MAX, "min": self.dist.ReduceOp.MIN,
                                    "sum": self.dist.ReduceOp.SUM}[op])
        return x.cpu().numpy()

    def allreduce_max(self, value: int) -> int:
        t = self.torch
        if self.use_host:
            x = t.tensor([int(value)], dtype=t.int64)
            self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX, group=self.host_group)
            return int(x.item())
        x = t.tensor([int(value)], dtype=t.int64, device="cpu" if self.staged else self.device)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX)
        return int(x.item())

    def allreduce_sum(self, values):
        t = self.torch
        x = t.tensor(np.asarray(values, dtype=np.float64), device="cpu" if self.staged else self.device)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
        return x.cpu().numpy()


class RankSim:
    """The Verlet step order of SURVEY.md section 3.1 on one rank of a decomposed run."""

    def __init__(self, ctx, pair, transport: Transport, grid, use_langevin=True, use_ucgstate=True, groupbit=1,
                 integrator="nve"):
        """integrator: "nve" = fix nve/ucgld, "wall" = fix nve/ucgld/wall/hard (Context.fix_nve_ucgld_wall_hard)"""
        self.ctx, self.pair, self.tr = ctx, pair, transport
        self.nve_kind = 2 if integrator == "wall" else 1
        self.density = getattr(pair, "style", "") == "table_ucg_bethe_density"
        self._aux_send = None
        self.cluster = False
        self.overlap = os.environ.get("UCG_HALO_OVERLAP", "0") == "1"  # measured: not a gain yet
        self.grid = list(grid)
        self.me = transport.rank
        self.world = transport.world
        self.use_langevin, self.use_ucgstate = use_langevin, use_ucgstate
        self.groupbit = groupbit
        self.atom_bytes, self.halo_bytes = ctx.record_bytes()
        ctx.decomp_set(self.grid, self.me)
        if transport.device.type == "cuda":
            # run the library's kernels on torch's current stream so that they are ordered with the
            # collectives (RCCL) and with torch's own copies (host-staged gloo)
            ctx.set_stream(transport.torch.cuda.current_stream().cuda_stream)
        self.ntimestep = self.beginstep = self.endstep = 0
        self.nrebuild = 0
        self.halo_send_counts = self.halo_recv_counts = None
        self._halo_send = None
        self._halo_plan = None

    def _buf(self, nbytes):
        t = self.tr.torch
        return t.empty(max(int(nbytes), 16), dtype=t.uint8, device=self.tr.device)

    def rebuild(self):
        ctx, tr = self.ctx, self.tr
        # exchange: every bead goes to the rank that owns its (wrapped) position
        sc = ctx.exchange_count()
        rc = tr.alltoall_counts(sc)
        sb = self._buf(sc.sum() * self.atom_bytes)
        ctx.exchange_pack(sb.data_ptr())
        rb = tr.alltoall_bytes(sb, sc, rc, self.atom_bytes)
        ctx.exchange_unpack(rb.data_ptr(), int(rc.sum()))
        # borders: periodic / neighbour-rank images inside each rank's extended brick
        sc = ctx.border_count()
        rc = tr.alltoall_counts(sc)
        sb = self._buf(sc.sum() * self.halo_bytes)
        ctx.border_pack(sb.data_ptr())
        rb = tr.alltoall_bytes(sb, sc, rc, self.halo_bytes)
        ctx.border_unpack(rb.data_ptr(), int(rc.sum()))
        self.halo_send_counts, self.halo_recv_counts = sc, rc
        self._halo_send = sb  # reused every step: same counts until the next rebuild
        self._keep = rb
        if self.cluster:
            # ghosts' group bits and molecule ids, for fix cluster_switch
            mb = self._buf(8 * int(sc.sum()))
            ctx.halo_molmask_pack(mb.data_ptr())
            rb2 = tr.alltoall_bytes(mb, sc, rc, 8)
            ctx.halo_molmask_unpack(rb2.data_ptr())
        self.nrebuild += 1

    # ---- fix cluster_switch across ranks: the reductions the reference does with MPI_Allreduce
    def cluster_switch(self, mol_seed, mol_offset, cutoff, seed, switch_freq, rate_file, contact_file):
        """fix ID all cluster_switch ... (every rank calls this after its beads and molecule ids are uploaded)"""
        ctx, tr = self.ctx, self.tr
        ctx.md_set_timestep(self.ntimestep)
        ctx.fix_cluster_switch(mol_seed, mol_offset, cutoff, seed, switch_freq, rate_file, contact_file, self.groupbit)
        s = ctx.cs_scalars()
        ctx.cs_set_scalars(tr.allreduce_array([s[0]], "max")[0], tr.allreduce_array([s[1]], "sum")[0],
                           tr.allreduce_array([s[2]], "sum")[0])
        for which in (1, 2, 4):  # mol_state, mol_restrict, presence
            ctx.cs_set_array(which, tr.allreduce_array(ctx.cs_array(which), "max"))
        self.cluster = True

    def _cluster_step(self):
        """check_cluster + attempt_switch on fresh lists (UCG/fix_cluster_switch.cpp:452-469)"""
        ctx, tr = self.ctx, self.tr
        changed = ctx.cs_sweep(1)
        while True:
            ctx.cs_set_array(5, tr.allreduce_array(ctx.cs_array(5), "min"))
            if not tr.allreduce_max(changed):
                break
            changed = ctx.cs_sweep(0)
        ctx.cs_finalize()
        ctx.cs_attempt_local()
        ctx.cs_set_array(3, tr.allreduce_array(ctx.cs_array(3), "max"))
        ctx.cs_attempt_apply()
        ctx.cs_advance()
        self.halo_forward()  # the ghosts' new atom types

    def halo_forward(self):
        ctx, tr = self.ctx, self.tr
        ctx.halo_pack(self._halo_send.data_ptr())
        if tr.staged:
            rb = tr.alltoall_bytes(self._halo_send, self.halo_send_counts, self.halo_recv_counts, self.halo_bytes)
        else:
            # same counts until the next rebuild: the split lists and the receive buffer are made once per rebuild
            # (the unpack of a step is ordered before the next step's collective on the stream, so one buffer does)
            plan = self._halo_plan
            if plan is None or plan[0] != self.nrebuild:
                ins = [int(c) * self.halo_bytes for c in self.halo_send_counts]
                outs = [int(c) * self.halo_bytes for c in self.halo_recv_counts]
                rbuf = tr.torch.empty(max(sum(outs), 1), dtype=tr.torch.uint8, device=tr.device)
                plan = self._halo_plan = (self.nrebuild, ins, outs, rbuf, self._halo_send[: sum(ins)], rbuf[: sum(outs)])
            tr.dist.all_to_all_single(plan[5], plan[4], plan[2], plan[1])
            rb = plan[3]
        ctx.halo_unpack(rb.data_ptr())
        self._keep = rb

    def halo_forward_and_pair(self):
        """a step without re-neighbouring and without energy output: the halo travels while the workgroups
        that touch no ghost are computed (ucg_pair_compute_part 1), the rest follows the unpack (part 2)"""
        ctx, tr = self.ctx, self.tr
        ctx.halo_pack(self._halo_send.data_ptr())
        work, rb = tr.alltoall_bytes_begin(self._halo_send, self.halo_send_counts, self.halo_recv_counts, self.halo_bytes)
        self.pair.compute_part(1)
        if work is not None:
            work.wait()
        ctx.halo_unpack(rb.data_ptr())
        self.pair.compute_part(2)
        self._keep = rb

    def _aux_halo(self, which):
        """forward one double2 per ghost (the density style's priors / CV forces) from the owner ranks"""
        ctx, tr = self.ctx, self.tr
        if self._aux_send is None or self._aux_send.numel() < 16 * int(self.halo_send_counts.sum()):
            self._aux_send = self._buf(16 * int(self.halo_send_counts.sum()))
        field = self.pair.density_buffer(which)
        ctx.halo_aux_pack(field, self._aux_send.data_ptr())
        rb = tr.alltoall_bytes(self._aux_send, self.halo_send_counts, self.halo_recv_counts, 16)
        ctx.halo_aux_unpack(field, rb.data_ptr())
        self._keep_aux = rb

    def _pair_compute(self, ev):
        if not self.density:
            return self.pair.compute(ev, ev)
        # table_ucg_bethe_density: its two mid-compute halos cross ranks
        self.pair.density_phase(1, ev, ev)
        self._aux_halo(0)
        self.pair.density_phase(2, ev, ev)
        self._aux_halo(1)
        return self.pair.density_phase(3, ev, ev)

    def _forces_and_post_force(self, ev):
        out = self._pair_compute(ev)
        if self.use_langevin:
            self.ctx.fix_ucgld_langevin_post_force(self.ntimestep, self.beginstep, self.endstep, self.groupbit)
        if self.use_ucgstate:
            self.ctx.fix_ucgstate_post_force()
        return out

    def thermo(self, last=None, mass=None, mvv2e=1.0):
        """One all-reduce per output step: E_pair, virial[6] of the last energy evaluation (`last` =
        what setup() / run() returned on this rank), the state-1 population, sum of lambda, the kinetic energy of x and
        of lambda, the bead count.  Host arithmetic on downloaded arrays: output steps are rare."""
        a = self.ctx.atoms_download()
        n = a["nlocal"]
        m = np.ones(n) if mass is None else np.asarray(mass, float)[a["type"][:n]]
        ke = 0.5 * mvv2e * float(np.sum(m * np.sum(a["v"][:n] ** 2, axis=1)))
        kel = 0.5 * mvv2e * float(np.sum(a["ucgml"][:n] * a["ucgvl"][:n] ** 2))
        e, vir = (last[0], list(last[1])) if last is not None else (0.0, [0.0] * 6)
        tot = self.tr.allreduce_sum([e] + vir + [float(a["ucgstate"][:n].sum()), float(a["ucgl"][:n].sum()), ke, kel, float(n)])
        return dict(eng_vdwl=tot[0], virial=np.array(tot[1:7]), state1=tot[7], sum_lambda=tot[8], ke=tot[9], ke_lambda=tot[10],
                    natoms=int(tot[11]))

    def setup(self, nsteps, ntypes=2):
        self.beginstep = self.ntimestep
        self.endstep = self.ntimestep + nsteps
        self.rebuild()
        if self.use_langevin:
            # Fix_UCGLD_Langevin::init() reads atom->ucgml[1..ntypes] of the local bead order (App. B #5)
            ml = self.ctx.atoms_download()["ucgml"]
            pad = np.full(ntypes + 1, ml[0] if len(ml) else 1.0)
            pad[: min(len(ml), ntypes + 1)] = ml[: ntypes + 1]
            self.ctx.fix_ucgld_langevin_init(ntypes, pad)
        return self._forces_and_post_force(1)

    def run(self, nsteps, thermo_every=0):
        ctx = self.ctx
        last = None
        initial_done = False
        for s in range(nsteps):
            self.ntimestep += 1
            ev = 1 if (thermo_every > 0 and self.ntimestep % thermo_every == 0) else 0
            if not initial_done:
                if self.nve_kind == 2:
                    ctx.fix_nve_ucgld_wall_hard_initial_integrate(self.groupbit)
                else:
                    ctx.fix_nve_ucgld_initial_integrate(self.groupbit)
            if self.cluster:
                ctx.md_set_timestep(self.ntimestep)
            due, flag = ctx.decide_local()
            fuse_next = (not ev) and (s + 1 < nsteps)
            rebuilt = bool(due and self.tr.allreduce_max(flag))
            if rebuilt:
                self.rebuild()
                if self.cluster and ctx.cs_due()[1]:
                    self._cluster_step()
            elif self.overlap and not ev and not self.density:
                self.halo_forward_and_pair()
                ctx.md_post_fused(self.use_langevin, self.use_ucgstate, self.nve_kind, fuse_next, self.ntimestep,
                                  self.beginstep, self.endstep, self.groupbit)
                initial_done = fuse_next
                continue
            else:
                self.halo_forward()
            # pair force, then langevin -> ucgstate -> final_integrate (-> next initial_integrate): one launch (the
            # gather kernel's epilogue) where that applies, else two
            if fuse_next and not self.density and ctx.md_pair_post(self.pair, self.use_langevin, self.use_ucgstate,
                                                                     self.nve_kind, self.ntimestep, self.beginstep,
                                                                     self.endstep, self.groupbit):
                pass
            else:
                out = self._pair_compute(ev)
                if ev:
                    last = out
                ctx.md_post_fused(self.use_langevin, self.use_ucgstate, self.nve_kind, fuse_next, self.ntimestep,
                                  self.beginstep, self.endstep, self.groupbit)
            initial_done = fuse_next
        return last


def run_bench(args, deck, beads, cs, rank, world, device_index, dist, shared, make_pair, attach_fixes, apply_options):
    """bench.py's N > 1 leg: the same beads split over `world` ranks (strong scaling).  `shared`: fewer GPUs than
    ranks (a one-GPU box rehearsing the path): the ranks share the GPUs and the halo is host-staged over gloo."""
    import time

    import torch

    from . import capi

    dt = 0.002
    device = torch.device("cuda", device_index)
    grid = choose_procgrid(world)
    # any initial split works: the first exchange sends every bead to its owner
    sl = slice(rank * beads.n // world, (rank + 1) * beads.n // world)
    ctx = capi.Context(device_index, dt=dt)
    n = sl.stop - sl.start
    ctx.atoms_upload(n, 0, beads.ntypes, beads.x[sl], beads.v[sl], beads.type[sl], beads.tag[sl], beads.mask[sl],
                     beads.ucgstate[sl], beads.ucgl[sl], beads.ucgvl[sl], beads.ucgml[sl], beads.ucgp[sl], beads.mass)
    if cs:
        ctx.upload_molecule(beads.molecule[sl])
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
    # lanes per bead: 0 = chosen from the rank's bead count so that a 1/8 brick still fills 256 CUs
    apply_options(ctx)
    pair = make_pair(ctx)
    use_lang, use_st = attach_fixes(ctx)
    tr = Transport(dist, device, staged=shared)
    wall = getattr(args, "integrator", "wall") == "wall"
    sim = RankSim(ctx, pair, tr, grid, use_langevin=use_lang, use_ucgstate=use_st, integrator="wall" if wall else "nve")
    if cs:
        sim.cluster_switch(cs["mol_seed"], 0, cs["cutoff"], cs["seed"], cs["switch_freq"], cs["rates"], cs["contacts"])
    sim.setup(args.warmup + args.steps, ntypes=beads.ntypes)
    sim.run(args.warmup)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    nre0 = sim.nrebuild
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.run(args.steps)
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    launches, pair_ms = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    pair.check_errors()
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    info = ctx.md_info()
    tot = tr.allreduce_sum([info["list_entries"], info["nghost"], info["nlocal"]])
    launches = args.steps  # the launches of one step (two parts, or the density style's three passes) count as one evaluation
    out = dict(elapsed=float(t.item()), n=beads.n, pair_launches=launches, pair_ms=pair_ms, list_entries=int(tot[0]),
               nghost=int(tot[1]), rebuilds=sim.nrebuild - nre0, maxrow=info["maxrow"], grid=grid,
               nlocal_sum=int(tot[2]), rank0_list_entries=info["list_entries"], rank0_nlocal=info["nlocal"],
               transport=(f"gloo, host-staged: {world} ranks share {torch.cuda.device_count()} GPU(s) (rehearsal of the N > 1 path)"
                          if shared else "RCCL (torch.distributed nccl backend) over xGMI"),
               small_messages=dict(route="gloo side group" if tr.use_host else "main backend", timed_us=tr.small_msg_us))
    if cs:
        out["cluster_switch_vector"] = [float(v) for v in ctx.fix_cluster_switch_vector()]
    return out
