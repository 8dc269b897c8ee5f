"""The built-in RCCL transport of the decomposed step loop (csrc/ucg_comm.hip: ucg_comm_attach_rccl, the grouped
ncclSend / ncclRecv all-to-all, the 8-byte count exchange, both ncclAllReduce wrappers) under the driver's pytest.

A test box has ONE GPU and RCCL refuses two ranks per device, so the transport runs with one rank: a 1 x 1 x 1
"decomposition" whose exchange / border / halo messages are all self-sends -- every call of the C++ loop's RCCL path
executes (ncclCommInitRank, grouped send + recv, ncclAllReduce SUM / MAX / MIN on int64 and f64), only the wire is
missing.  Each run must give the SAME BITS as the same run on the callback communicator (here: in-process callbacks
that copy device to device), the transport every 2- and 4-rank test uses -- and as the ORACLE's single-rank run, whose
ucg-rebuild-v1 order a one-rank decomposition reproduces."""
import ctypes as C

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


class LocalComm:
    """ucg_comm_ops for world = 1: the block a rank sends to itself is copied device to device"""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.calls = dict(alltoallv=0, alltoall_ll=0, allreduce_ll=0, allreduce_f64=0)

    def alltoallv(self, user, send, sendbytes, recv, recvbytes, stream):
        self.calls["alltoallv"] += 1
        if sendbytes[0] != recvbytes[0]:
            return 1
        # ordered on the context's stream (a plain hipMemcpy between device buffers runs on the null stream and does not
        # wait for, or hold back, a non-blocking stream: the unpack kernel could read the buffer before the copy landed)
        if sendbytes[0] and self.hip.hipMemcpyAsync(recv, send, sendbytes[0], 3, stream):  # hipMemcpyDeviceToDevice
            return 1
        return 1 if self.hip.hipStreamSynchronize(stream) else 0

    def alltoall_ll(self, user, send, recv):
        self.calls["alltoall_ll"] += 1
        recv[0] = send[0]
        return 0

    def allreduce_ll(self, user, buf, n, op):
        self.calls["allreduce_ll"] += 1
        return 0

    def allreduce_f64(self, user, buf, n, op):
        self.calls["allreduce_f64"] += 1
        return 0


def _run(pkg, transport, case, beads, deck, steps, every, dt):
    ctx = pkg.capi.Context(-1, dt=dt)
    try:
        ctx.upload_beads(beads)
        if case == "cluster":
            ctx.upload_molecule(beads.molecule)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=every, delay=0, check=1)
        ctx.decomp_set([1, 1, 1], 0)
        lc = None
        if transport == "rccl":
            # (ncclCommInitRank has been seen to fail once with "unhandled cuda error" on a box right after other tests'
            # rank processes had exited; a failed attach leaves no state behind, so a one-rank communicator is simply asked
            # for again -- with a fresh id -- before the failure counts)
            for attempt in range(3):
                try:
                    ctx.comm_attach_rccl(pkg.capi.Context.rccl_unique_id(), 0, 1)
                    break
                except pkg.capi.UcgError as e:
                    if "ncclCommInitRank" not in str(e) or attempt == 2:
                        raise
                    import time
                    time.sleep(3.0)
            assert ctx.comm_info() == dict(rank=0, world=1, rccl=True, nrebuild=0)
            assert ctx.comm_allreduce_sum([1.0, 2.5])[1] == 2.5
        elif transport == "callbacks":
            lc = LocalComm()
            ctx.comm_attach(0, 1, lc.alltoallv, lc.alltoall_ll, lc.allreduce_ll, lc.allreduce_f64)
        lang = ust = False
        if case == "ucgld":
            gp = util.gpu_pair(ctx, "table_ucgld", deck)
            ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279)
            ctx.fix_ucgstate("mc", 9127, 0.3)
            ctx.fix_nve_ucgld_wall_hard(False, 0.1)
            lang = ust = True
            nve = "wall"
        elif case == "density":
            gp = util.gpu_pair(ctx, "table_ucg_bethe_density", deck)
            ctx.fix_ucgstate("mc", 4242, 0.3)
            ust, nve = True, True
        else:
            gp = util.gpu_pair_multi(ctx, "table_ucgld", deck)
            nve = True
        ctx.md_attach(gp, nve=nve, langevin=lang, ucgstate=ust)
        if case == "cluster":
            ctx.fix_cluster_switch(*deck.cluster_args)
        ctx.md_setup(steps)
        th0 = ctx.md_thermo()
        ctx.md_run(steps, steps)
        gp.check_errors()
        out = ctx.atoms_download()
        out["thermo0"], out["thermo1"], out["info"] = th0, ctx.md_thermo(), ctx.md_info()
        if transport != "resident":
            out["comm"] = ctx.comm_info()
        if case == "cluster":
            out["cs"] = ctx.fix_cluster_switch_arrays()
            out["vec"] = ctx.fix_cluster_switch_vector()
            out["mol"] = ctx.download_molecule()
        if lc:
            out["calls"] = lc.calls
        gp.close()
        return out
    finally:
        ctx.close()


def _same_bits(a, b, keys=("x", "v", "f", "ucgl", "ucgvl", "ucgforce", "scores", "ucgp")):
    assert np.array_equal(a["tag"], b["tag"]) and np.array_equal(a["ucgstate"], b["ucgstate"]) and np.array_equal(a["type"], b["type"])
    for k in keys:
        assert util.bits_equal(a[k], b[k]), k


@pytest.mark.parametrize("self_send", ["1", "0"])
@pytest.mark.parametrize("case", ["ucgld", "density", "cluster"])
def test_one_rank_on_rccl_equals_the_callback_transport_and_the_oracle(pkg, orc, case, self_send, monkeypatch):
    # "1": the block a rank sends to itself goes through ncclSend / ncclRecv like a peer's (the grouped path every
    # multi-rank run takes); "0", the default: it is a device copy and only counts and reductions use RCCL
    monkeypatch.setenv("UCG_RCCL_SELF_SEND", self_send)
    steps, every = 40, 2
    if case == "ucgld":
        dt = 0.004
        deck = util.make_deck("spline", 1024)
        beads = pkg.synth.make_beads(10, seed=5)
        op = util.oracle_pair("table_ucgld", deck)
        sim = util.oracle_sim(beads, op, mode=1, dt=dt, langevin=(1.0, 1.0, 1.0, 48279), nve="wall", ucgstate=("mc", 9127, 0.3), every=every)
    elif case == "density":
        dt = 0.002
        deck = util.make_deck("spline", 1024, density=(11.3, 1.5), extra11=0.05)
        beads = pkg.synth.make_beads(10, seed=5)
        op = util.oracle_pair("table_ucg_bethe_density", deck)
        sim = util.oracle_sim(beads, op, mode=1, dt=dt, nve=True, ucgstate=("mc", 4242, 0.3), every=every)
    else:
        dt, every = 0.004, 5
        deck = util.make_multi_deck(2, "spline", 256)
        beads = util.multi_type_beads(pkg, 10, 2, seed=5, molecule_size=2)
        rates, contacts = pkg.synth.write_cluster_switch_files(deck.workdir, 0.35, [1], [2], [(1, 1)])
        mol_seed = int(beads.molecule[np.flatnonzero(beads.type == 1)[0]])
        deck.cluster_args = (mol_seed, 0, 1.15, 4711, 5, rates, contacts)
        op = util.oracle_pair_multi("table_ucgld", deck)
        sim = util.oracle_sim(beads, op, mode=1, dt=dt, nve=True, every=every)
        sim.cluster_switch(*deck.cluster_args)
    R = _run(pkg, "rccl", case, beads, deck, steps, every, dt)
    K = _run(pkg, "callbacks", case, beads, deck, steps, every, dt)
    assert R["comm"]["rccl"] and not K["comm"]["rccl"]
    assert R["comm"]["nrebuild"] == K["comm"]["nrebuild"] >= 3
    assert R["nghost"] == K["nghost"] > 0
    _same_bits(R, K)
    for k in ("thermo0", "thermo1"):
        assert R[k]["eng_vdwl"] == K[k]["eng_vdwl"] and np.array_equal(R[k]["virial"], K[k]["virial"])
    # the callback run did go through the communicator: exchanges + borders at the rebuilds, one halo per step
    # (the density style: two more per force evaluation), the re-neighbour flag every `every` steps
    assert K["calls"]["alltoallv"] >= steps and K["calls"]["alltoall_ll"] >= 2 * K["comm"]["nrebuild"]
    assert K["calls"]["allreduce_ll"] >= steps // every and K["calls"]["allreduce_f64"] >= 2
    if case == "cluster":
        for k in ("mol_cluster", "mol_state", "mol_restrict", "mol_accept"):
            assert np.array_equal(R["cs"][k], K["cs"][k]), k
        assert np.array_equal(R["vec"], K["vec"]) and R["vec"][0] > 50 and 0 < R["vec"][1] < R["vec"][0]
        assert np.array_equal(R["mol"], K["mol"])
        assert (R["type"] != beads.type[R["tag"] - 1]).sum() > 0  # switching happened
    # ... and the resident single-rank loop of the library and the oracle
    S = _run(pkg, "resident", case, beads, deck, steps, every, dt)
    _same_bits(R, S)
    assert sim.setup(steps) == 0 and sim.run(steps, steps) == 0
    O = sim.arrays()
    _same_bits(R, O)
    assert abs(R["thermo1"]["eng_vdwl"] - sim.ev()["eng_vdwl"]) <= 1e-12 * abs(sim.ev()["eng_vdwl"])
    if case != "cluster":
        assert 0 < R["ucgstate"].sum() < beads.n


def test_rccl_load_failure_is_an_error_code_not_a_crash(pkg, monkeypatch):
    """a librccl that fails to resolve must leave the loader unloaded (UCG_ERR_COMM on every later call)"""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import conftest; pkg = conftest.load_package(); C = pkg.capi\n"
            "for _ in range(2):\n"
            "    try:\n"
            "        C.Context.rccl_unique_id(); print('loaded')\n"
            "    except C.UcgError as e:\n"
            "        print('err', e.code)\n" % os.path.dirname(os.path.abspath(__file__)))
    # a shared object that is NOT rccl: dlopen succeeds, the first dlsym fails
    env = dict(os.environ, UCG_RCCL_LIBRARY="libm.so.6")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["err", "8", "err", "8"], r.stdout
