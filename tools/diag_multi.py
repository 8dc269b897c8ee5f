#!/usr/bin/env python3
"""per-phase wall time of the decomposed step loop with ONE rank on the RCCL transport
(each phase followed by a device synchronize, so the sum is an upper bound of a pipelined step)"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pkg = entry.load_package()
capi, synth, multi = pkg.capi, pkg.synth, pkg.multi
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 50
beads = synth.make_beads(ncell, seed=1)
deck = synth.make_deck(tempfile.mkdtemp(), "spline", 1024)
ctx = capi.Context(0, dt=0.002)
ctx.upload_beads(beads)
ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
ctx.set_option("gather_slots", 0)
pair = capi.Pair(ctx, "table_ucgld")
pair.settings(deck.pair_style_args()); pair.coeff(deck.pair_coeff_args()); pair.init(2, 1.0)
ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279)
ctx.fix_ucgstate("ld")
tr = multi.Transport(dist, torch.device("cuda", 0), staged=False)
sim = multi.RankSim(ctx, pair, tr, [1, 1, 1])
sim.setup(1000)
sim.run(20)
torch.cuda.synchronize()
acc = {}
def timed(name, fn):
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
N = 100
t_all = time.perf_counter()
for s in range(N):
    sim.ntimestep += 1
    timed("initial", lambda: ctx.fix_nve_ucgld_initial_integrate(1))
    due, flag = timed("decide", lambda: ctx.decide_local())
    if due and timed("allreduce", lambda: tr.allreduce_max(flag)):
        timed("rebuild", sim.rebuild)
    else:
        timed("halo_pack", lambda: ctx.halo_pack(sim._halo_send.data_ptr()))
        rb = timed("alltoall", lambda: tr.alltoall_bytes(sim._halo_send, sim.halo_send_counts, sim.halo_recv_counts, sim.halo_bytes))
        timed("halo_unpack", lambda: ctx.halo_unpack(rb.data_ptr()))
    timed("pair", lambda: pair.compute(0, 0))
    timed("post_fused", lambda: ctx.md_post_fused(True, True, True, False, sim.ntimestep, sim.beginstep, sim.endstep))
# rebuild phases, one by one
reb = {}
def rt(name, fn):
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); reb[name] = reb.get(name, 0.0) + time.perf_counter() - t0; return r
for rep in range(5):
    sc = rt("exchange_count", ctx.exchange_count)
    rc = rt("counts_a2a", lambda: tr.alltoall_counts(sc))
    sb = rt("alloc", lambda: sim._buf(sc.sum() * sim.atom_bytes))
    rt("exchange_pack", lambda: ctx.exchange_pack(sb.data_ptr()))
    rb = rt("exchange_a2a", lambda: tr.alltoall_bytes(sb, sc, rc, sim.atom_bytes))
    rt("exchange_unpack", lambda: ctx.exchange_unpack(rb.data_ptr(), int(rc.sum())))
    sc = rt("border_count(sort)", ctx.border_count)
    rc = rt("counts_a2a", lambda: tr.alltoall_counts(sc))
    sb = rt("alloc", lambda: sim._buf(sc.sum() * sim.halo_bytes))
    rt("border_pack", lambda: ctx.border_pack(sb.data_ptr()))
    rb = rt("border_a2a", lambda: tr.alltoall_bytes(sb, sc, rc, sim.halo_bytes))
    rt("border_unpack(rows)", lambda: ctx.border_unpack(rb.data_ptr(), int(rc.sum())))
    sim.halo_send_counts, sim.halo_recv_counts, sim._halo_send, sim._keep = sc, rc, sb, rb
print("rebuild phases (us each):")
for k, v in reb.items():
    print(f"  {k:22s} {v / 5 * 1e6:9.1f}")
print("beads", beads.n, "wall per step (sync after each phase): %.1f us" % ((time.perf_counter() - t_all) / N * 1e6))
for k, v in acc.items():
    print(f"  {k:12s} {v / N * 1e6:9.1f} us/step")
# host cost of each call alone (no sync inside the step; one sync per step keeps the queue empty)
host = {}
def ht(name, fn):
    t0 = time.perf_counter(); r = fn(); host[name] = host.get(name, 0.0) + time.perf_counter() - t0; return r
nh = 0
for s in range(N):
    sim.ntimestep += 1
    ht("initial", lambda: ctx.fix_nve_ucgld_initial_integrate(1))
    due, flag = ht("decide", lambda: ctx.decide_local())
    if due and tr.allreduce_max(flag):
        sim.rebuild()
    else:
        nh += 1
        ht("halo_pack", lambda: ctx.halo_pack(sim._halo_send.data_ptr()))
        rb = ht("alltoall", lambda: tr.alltoall_bytes(sim._halo_send, sim.halo_send_counts, sim.halo_recv_counts, sim.halo_bytes))
        ht("halo_unpack", lambda: ctx.halo_unpack(rb.data_ptr()))
    ht("pair", lambda: pair.compute(0, 0))
    ht("post_fused", lambda: ctx.md_post_fused(True, True, True, False, sim.ntimestep, sim.beginstep, sim.endstep))
    torch.cuda.synchronize()
print("host time of each call (no sync), us/step:")
for k, v in host.items():
    print(f"  {k:12s} {v / N * 1e6:9.1f}")
torch.cuda.synchronize(); t0 = time.perf_counter(); sim.run(N); torch.cuda.synchronize()
print("pipelined sim.run: %.1f us/step" % ((time.perf_counter() - t0) / N * 1e6))
dist.destroy_process_group()
