"""Spatial decomposition across the GPUs of one node: one process per GPU.

Replaces, for the resident path, what upstream LAMMPS' CommBrick does with the field lists of
``UCG/atom_vec_ucg.cpp:66-82``: bead migration (``exchange``), ghost construction
(``borders``) and the per-step forward halo (``forward_comm``: x, ucgstate, ucgl, ucgp).
The gather kernels accumulate nothing on ghosts, so there is NO reverse halo.

Device work (count / pack / unpack, sorting, binning, lists, forces, fixes) AND the rank-level
step loop are the library's (csrc/ucg_comm.hip: ucg_md_setup / ucg_md_run with a communicator
attached).  On a multi-GPU node the communicator is RCCL, called directly from that C++ loop
(grouped ncclSend / ncclRecv to the <= 7 peers: direct neighbour exchange on the point-to-point
xGMI links, not LAMMPS' staged x/y/z forwarding).  This module only attaches the communicator
and, for ranks that share a GPU (tests, rehearsals), provides the callback form of it over
``torch.distributed`` (gloo) with host staging.
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


class HostStagedComm:
    """The callbacks of include/ucg_hip.h's ucg_comm_ops over torch.distributed (gloo), with the device buffers staged
    through the host: for ranks that SHARE a GPU (tests, rehearsals on a one-GPU box), where RCCL cannot be used.  A
    LAMMPS build would put MPI calls in the same four callbacks."""

    def __init__(self, dist, world, rank):
        import ctypes as C

        import torch

        self.C, self.torch, self.dist, self.world, self.rank = C, torch, dist, world, rank
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.failed = None

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as exc:  # noqa: BLE001  (an exception must not cross the C ABI)
            self.failed = exc
            return 1

    def alltoallv(self, user, send, sendbytes, recv, recvbytes, stream):
        def go():
            t, C = self.torch, self.C
            sb = [int(sendbytes[r]) for r in range(self.world)]
            rb = [int(recvbytes[r]) for r in range(self.world)]
            if self.hip.hipStreamSynchronize(stream):
                raise RuntimeError("hipStreamSynchronize failed")
            hs = t.empty(max(sum(sb), 1), dtype=t.uint8)
            hr = t.empty(max(sum(rb), 1), dtype=t.uint8)
            if sum(sb) and self.hip.hipMemcpy(hs.data_ptr(), send, sum(sb), 2):  # hipMemcpyDeviceToHost
                raise RuntimeError("hipMemcpy d2h failed")
            self.dist.all_to_all_single(hr[: sum(rb)], hs[: sum(sb)], rb, sb)
            if sum(rb) and self.hip.hipMemcpy(recv, hr.data_ptr(), sum(rb), 1):  # hipMemcpyHostToDevice
                raise RuntimeError("hipMemcpy h2d failed")
        return self._guard(go)

    def alltoall_ll(self, user, send, recv):
        def go():
            t = self.torch
            s = t.tensor([int(send[r]) for r in range(self.world)], dtype=t.int64)
            r = t.empty_like(s)
            self.dist.all_to_all_single(r, s)
            for i in range(self.world):
                recv[i] = int(r[i])
        return self._guard(go)

    def _allreduce(self, buf, n, op, dtype):
        t = self.torch
        x = t.tensor([buf[i] for i in range(n)], dtype=dtype)
        self.dist.all_reduce(x, op=(self.dist.ReduceOp.SUM, self.dist.ReduceOp.MAX, self.dist.ReduceOp.MIN)[op])
        for i in range(n):
            buf[i] = x[i].item()

    def allreduce_ll(self, user, buf, n, op):
        return self._guard(lambda: self._allreduce(buf, n, op, self.torch.int64))

    def allreduce_f64(self, user, buf, n, op):
        return self._guard(lambda: self._allreduce(buf, n, op, self.torch.float64))


class RankSim:
    """One rank of a decomposed run.  The rank-level Verlet loop (SURVEY.md section 3.1 with CommBrick's exchange /
    borders / forward_comm and the MPI_Allreduce steps around it) runs INSIDE the library (csrc/ucg_comm.hip, ucg_md_setup
    / ucg_md_run with a communicator attached); this class only attaches the communicator:
      * `rccl_id` given: the built-in RCCL transport (one process per GPU, grouped ncclSend / ncclRecv over xGMI);
      * else: torch.distributed (gloo) callbacks with host staging -- ranks sharing one GPU."""

    def __init__(self, ctx, pair, transport: Transport, grid, use_langevin=True, use_ucgstate=True, groupbit=1,
                 integrator="nve", rccl_id=None):
        """integrator: "nve" = fix nve/ucgld, "wall" = fix nve/ucgld/wall/hard (Context.fix_nve_ucgld_wall_hard)"""
        self.ctx, self.pair, self.tr = ctx, pair, transport
        self.grid = list(grid)
        self.me, self.world = transport.rank, transport.world
        self.use_langevin, self.use_ucgstate = use_langevin, use_ucgstate
        ctx.decomp_set(self.grid, self.me)
        self.host_comm = None
        self.transport_note = None
        if rccl_id is not None:
            # every rank reports whether RCCL came up (and a first all-reduce gives the rank count); if any did not,
            # ALL ranks fall back to the host-staged callbacks together, and the caller is told (transport_note)
            why = None
            try:
                ctx.comm_attach_rccl(rccl_id, self.me, self.world)
                if int(round(float(ctx.comm_allreduce_sum([1.0])[0]))) != self.world:
                    why = "first ncclAllReduce returned a wrong rank count"
            except Exception as e:  # noqa: BLE001 -- reported below, by every rank
                why = f"{type(e).__name__}: {e}"
            reasons = [None] * self.world
            transport.dist.all_gather_object(reasons, why)
            bad = [f"rank {r}: {w}" for r, w in enumerate(reasons) if w]
            if bad:
                try:
                    ctx.comm_detach()
                except Exception:  # noqa: BLE001
                    pass
                rccl_id = None
                self.transport_note = "RCCL transport unavailable (" + "; ".join(bad[:2]) + "): host-staged gloo callbacks used instead"
        if rccl_id is None:
            self.host_comm = HostStagedComm(transport.dist, self.world, self.me)
            ctx.comm_attach(self.me, self.world, self.host_comm.alltoallv, self.host_comm.alltoall_ll,
                            self.host_comm.allreduce_ll, self.host_comm.allreduce_f64)
        ctx.md_attach(pair, nve="wall" if integrator == "wall" else True, langevin=use_langevin, ucgstate=use_ucgstate)

    @property
    def nrebuild(self):
        return self.ctx.comm_info()["nrebuild"]

    def _chk(self, fn, *a):
        try:
            return fn(*a)
        except Exception:
            if self.host_comm is not None and self.host_comm.failed is not None:
                raise self.host_comm.failed
            raise

    def cluster_switch(self, mol_seed, mol_offset, cutoff, seed, switch_freq, rate_file, contact_file):
        """fix ID all cluster_switch ... (every rank, after its beads and molecule ids are uploaded); the reductions the
        reference does with MPI_Allreduce at creation are made by the library at the next setup()"""
        self.ctx.fix_cluster_switch(mol_seed, mol_offset, cutoff, seed, switch_freq, rate_file, contact_file)

    def _last(self):
        th = self.ctx.md_thermo()
        return th["eng_vdwl"], th["virial"]

    def setup(self, nsteps, ntypes=2):
        self._chk(self.ctx.md_setup, nsteps)
        return self._last()  # totals over the ranks

    def run(self, nsteps, thermo_every=0):
        self._chk(self.ctx.md_run, nsteps, thermo_every)
        return self._last() if thermo_every > 0 else None

    def thermo(self, last=None, mass=None, mvv2e=1.0):
        """One all-reduce per output step: E_pair, virial[6] of the last energy evaluation (already totals:
        `last` = what setup() / run() returned), the state-1 population, sum of lambda, the kinetic energy of x and of
        lambda, the bead count.  Host arithmetic on downloaded arrays: output steps are rare."""
        a = self.ctx.atoms_download()
        n = a["nlocal"]
        m = np.ones(n) if mass is None else np.asarray(mass, float)[a["type"][:n]]
        ke = 0.5 * mvv2e * float(np.sum(m * np.sum(a["v"][:n] ** 2, axis=1)))
        kel = 0.5 * mvv2e * float(np.sum(a["ucgml"][:n] * a["ucgvl"][:n] ** 2))
        e, vir = (last[0], list(last[1])) if last is not None else (0.0, [0.0] * 6)
        tot = self.ctx.comm_allreduce_sum([float(a["ucgstate"][:n].sum()), float(a["ucgl"][:n].sum()), ke, kel, float(n)])
        return dict(eng_vdwl=e, virial=np.array(vir), state1=tot[0], sum_lambda=tot[1], ke=tot[2], ke_lambda=tot[3],
                    natoms=int(tot[4]))


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
    tr = Transport(dist, device, staged=True)  # torch.distributed (gloo): rendezvous, barriers, host-side sums only
    wall = getattr(args, "integrator", "wall") == "wall"
    rccl_id = rccl_why = None
    if not shared:  # one GPU per rank: the library's RCCL transport; rank 0's id travels over the gloo group
        box = [None, None]
        if rank == 0:
            try:
                box[0] = capi.Context.rccl_unique_id()
            except Exception as e:  # noqa: BLE001 -- librccl missing: every rank takes the host-staged transport
                box[1] = f"{type(e).__name__}: {e}"
        dist.broadcast_object_list(box, src=0)
        rccl_id, rccl_why = box
    sim = RankSim(ctx, pair, tr, grid, use_langevin=use_lang, use_ucgstate=use_st, integrator="wall" if wall else "nve",
                  rccl_id=rccl_id)
    if cs:
        sim.cluster_switch(cs["mol_seed"], 0, cs["cutoff"], cs["seed"], cs["switch_freq"], cs["rates"], cs["contacts"])
    equil = getattr(args, "equilibrate", 0)
    sim.setup(equil + args.warmup + args.steps, ntypes=beads.ntypes)
    sim.run(equil + args.warmup)  # input preparation (melt the lattice), then the warm-up
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
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    info = ctx.md_info()
    tot = tr.allreduce_sum([info["list_entries"], info["nghost"], info["nlocal"]])
    launches = args.steps  # the launches of one step (two parts, or the density style's three passes) count as one evaluation
    out = dict(elapsed=float(t.item()), n=beads.n, pair_launches=launches, pair_ms=pair_ms, list_entries=int(tot[0]),
               nghost=int(tot[1]), rebuilds=sim.nrebuild - nre0, maxrow=info["maxrow"], grid=grid,
               nlocal_sum=int(tot[2]), rank0_list_entries=info["list_entries"], rank0_nlocal=info["nlocal"],
               transport=(f"gloo callbacks, host-staged: {world} ranks share {torch.cuda.device_count()} GPU(s) (rehearsal of the N > 1 path)"
                          if shared else "RCCL called from the library's C++ step loop (grouped ncclSend / ncclRecv over xGMI)"))
    tinfo = ctx.comm_transport()
    out["rccl"] = tinfo["rccl"]
    out["rccl_nranks"] = tinfo["rccl_nranks"]  # ncclCommCount of the attached communicator (0: no RCCL)
    if not shared and not tinfo["rccl"]:
        out["transport"] = sim.transport_note or f"RCCL transport unavailable ({rccl_why}): host-staged gloo callbacks used instead"
        # every rank has a GPU of its own and RCCL still did not attach: a scaling number measured over host-staged gloo
        # must not pass for an xGMI one.  UCG_BENCH_ALLOW_HOST_STAGED=1 lets the run stand (its JSON says "rccl": false).
        if os.environ.get("UCG_BENCH_ALLOW_HOST_STAGED") != "1":
            raise SystemExit("bench.py: --gpus %d found a GPU per rank but RCCL did not attach (%s); refusing to report a "
                             "host-staged number as the scaling result (UCG_BENCH_ALLOW_HOST_STAGED=1 overrides)" % (world, out["transport"]))
    if tinfo["rccl"] and tinfo["rccl_nranks"] != world:
        raise SystemExit(f"bench.py: RCCL reports {tinfo['rccl_nranks']} ranks for a world of {world}")
    if cs:
        out["cluster_switch_vector"] = [float(v) for v in ctx.fix_cluster_switch_vector()]
    return out
