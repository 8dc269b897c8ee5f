"""debug aid: which transport of the one-rank decomposed run leaves the oracle's trajectory, and at which step"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
import test_gpu_rccl as T
from conftest import load_package, load_oracle
pkg, orc = load_package(), load_oracle()
vrow = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps, every, dt = 40, 2, 0.004
deck = util.make_deck("spline", 1024)
beads = pkg.synth.make_beads(10, seed=5)
op = util.oracle_pair("table_ucgld", deck)
op.set_sum_fixed(bool(vrow))
sim = util.oracle_sim(beads, op, mode=1, dt=dt, langevin=(1.0, 1.0, 1.0, 48279), nve="wall", ucgstate=("mc", 9127, 0.3), every=every)
assert sim.setup(steps) == 0
ref = []
for s in range(steps // 2):
    assert sim.run(2, 0) == 0
    A = sim.arrays()
    o = np.argsort(A["tag"])
    ref.append((A["x"][o].copy(), A["f"][o].copy()))

def run(transport):
    ctx = pkg.capi.Context(-1, dt=dt)
    ctx.set_option("pair_vrow", vrow)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=every, delay=0, check=1)
    lc = None
    if transport != "resident":
        ctx.decomp_set([1, 1, 1], 0)
    if transport == "rccl":
        ctx.comm_attach_rccl(pkg.capi.Context.rccl_unique_id(), 0, 1)
    elif transport == "callbacks":
        lc = T.LocalComm()
        ctx.comm_attach(0, 1, lc.alltoallv, lc.alltoall_ll, lc.allreduce_ll, lc.allreduce_f64)
    gp = util._gpu_pair_raw(ctx, "table_ucgld", deck)
    ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279)
    ctx.fix_ucgstate("mc", 9127, 0.3)
    ctx.fix_nve_ucgld_wall_hard(False, 0.1)
    ctx.md_attach(gp, nve="wall", langevin=True, ucgstate=True)
    ctx.md_setup(steps)
    first = None
    for s in range(steps // 2):
        ctx.md_run(2, 0)
        G = ctx.atoms_download()
        o = np.argsort(G["tag"])
        okx = np.array_equal(G["x"][o].view(np.uint64), ref[s][0].view(np.uint64))
        okf = np.array_equal(G["f"][o].view(np.uint64), ref[s][1].view(np.uint64))
        if not (okx and okf) and first is None:
            bad = np.flatnonzero(np.any(G["f"][o] != ref[s][1], axis=1))
            first = (2 * (s + 1), okx, okf, len(bad), bad[:6].tolist(), ctx.md_info()["nrebuild"],
                     float(np.abs(G["f"][o] - ref[s][1]).max()))
    gp.check_errors()
    gp.close()
    ctx.close()
    return first

for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    for tr in ("rccl", "callbacks", "resident"):
        os.environ["UCG_RCCL_SELF_SEND"] = "1" if rep % 2 else "0"
        print(rep, tr, "first deviation:", run(tr), flush=True)
