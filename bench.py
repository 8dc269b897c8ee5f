#!/usr/bin/env python3
"""bench.py -- UCG hot path on MI355X: timesteps/s at 1 M UCG beads.

  python bench.py --gpus N --steps K --warmup W            (N > 1: this process starts the N ranks itself)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (same ranks)

One "step" is one full Verlet step of the resident path on synthetic input:
fix nve/ucgld/wall/hard initial_integrate -> re-neighbour decision (every 10 steps, rebuild when a
bead moved skin/2) / halo refresh -> pair_style table_ucgld (spline 1024, 2-state, the
north-star neighbour loop) -> fix ucgld/langevin -> fix ucgstate ld -> final_integrate.
(The hard-wall variant of the lambda integrator: plain fix nve/ucgld lets lambda drift without bound and the
melt breaks down after ~1000 steps, in the reference as here -- see --integrator; the JSON line carries a short
`fix nve/ucgld` leg beside the headline number.)
Workload (BASELINE.md config 4 at N GPUs, config-2 styles at 1 M beads): 100^3 beads at
rho* = 0.8, rc = 2.5, skin = 0.3, dt = 0.002, fp64 throughout.  For N > 1 the SAME beads
are split across the ranks (strong scaling).  Inputs are resident in HBM before timing.
`--config 2|3|4|5` selects the other configurations of BASELINE.json (5 = 4 M beads, table_ucg_bethe_density +
fix ucgstate mc + fix cluster_switch).

The JSON line also carries
  roofline     -- the pair kernel's algorithmic bytes (44 B per half-list entry + 96 B per bead;
                  density style 116 B per full-list entry + 220 B per bead) / its mean duration from HIP events
                  on its own stream, vs 8 TB/s; plus the fp64-issue bound that actually binds the kernel
  cpu_baseline -- the oracle's reference-order (half list, scalar) loop timed on this box's host
                  cores on a bounded sample; a reported baseline, not a target.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

# multi-process GPU work on this pool needs dmabuf IPC (RCCL across ranks); set before anything touches the GPU
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
SIMDS = 256 * 4        # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4        # max clock; a wave64 fp64 instruction holds its SIMD for 4 cycles (16 lanes / cycle)
MEASURED_CLOCK_GHZ = 2.0  # effective shader clock under the pair kernel (profiles/README.md, round 2: 1.92-2.10)

CLUSTER_SWITCH = dict(prob_on=0.35, cutoff=1.2, seed=4711, switch_freq=50, molecule_size=2)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (SURVEY.md 8d: >= 1000, re-neighbouring included)")
    ap.add_argument("--warmup", type=int, default=200, help="untimed steps directly before the timed ones")
    ap.add_argument("--equilibrate", type=int, default=200,
                    help="steps run on the path itself while PREPARING the synthetic input (SURVEY.md 8d: 'simple-cubic lattice ... "
                         "jitter ... then 200 warm-up steps on the CPU/GPU path itself'): the workload is the equilibrated melt, "
                         "not the jittered lattice (which has 10 %% more list entries); untimed, before --warmup")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="a configuration of BASELINE.json: 2 = 262 144 beads table_ucgld + langevin; 3 = 1 M beads "
                         "table_ucg_bethe; 4 = the default workload (1 M beads table_ucgld, decomposed for --gpus > 1); "
                         "5 = 4 M beads (fcc) table_ucg_bethe_density + fix ucgstate mc + fix cluster_switch")
    ap.add_argument("--ncell", type=int, default=None, help="beads = ncell^3 (default 100 -> 1 M; --config 2: 64)")
    ap.add_argument("--tabstyle", default="spline")
    ap.add_argument("--tablength", type=int, default=1024)
    ap.add_argument("--style", default="table_ucgld",
                    choices=["table_ucgld", "table_ucg_bethe", "table_ucg_bethe_density"],
                    help="pair style (default: the headline table_ucgld workload; the others are BASELINE.md configs 3 "
                         "and 5, reported with their own algorithmic bytes)")
    ap.add_argument("--lattice", default="sc", choices=["sc", "fcc"],
                    help="sc: ncell^3 beads (default); fcc: 4 ncell^3 beads (BASELINE.json's 4 M-bead configuration = --ncell 100 --lattice fcc)")
    ap.add_argument("--cluster-switch", action="store_true",
                    help="two actual atom types (ON / OFF), molecules of two beads, fix cluster_switch every "
                         f"{CLUSTER_SWITCH['switch_freq']} steps (BASELINE.json config 5)")
    ap.add_argument("--integrator", default="wall", choices=["wall", "nve"],
                    help="wall = fix nve/ucgld/wall/hard (default), nve = fix nve/ucgld.  The latter never clamps lambda "
                         "(UCG/fix_nve_ucgld.cpp:44-153): with the linear-in-lambda mixing nothing confines it, and after "
                         "~1000 steps of this melt it has drifted far enough outside [0, 1] for the mixed potentials to "
                         "turn attractive at contact -- the reference stops there with 'Pair distance < table inner "
                         "cutoff' (tools/stability.py).  The hard-wall variant of the same integrator is stable.")
    ap.add_argument("--no-nve-leg", action="store_true", help="skip the short `fix nve/ucgld` leg printed beside the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropin-steps", type=int, default=300,
                    help="steps of the drop-in leg printed beside the headline (N = 1): the same workload hook by hook in upstream "
                         "Verlet's order through the C ABI with the caller's (pinned) arrays bound as lazily synchronised host "
                         "mirrors; 0 = skip")
    ap.add_argument("--cpu-ncell", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the multi-core baseline (0 = min(cores, 16))")
    args = ap.parse_args(argv)
    preset_ncell = 100
    if args.config == 2:
        preset_ncell, args.style, args.lattice = 64, "table_ucgld", "sc"
    elif args.config == 3:
        args.style, args.lattice = "table_ucg_bethe", "sc"
    elif args.config == 4:
        args.style, args.lattice = "table_ucgld", "sc"
    elif args.config == 5:
        args.style, args.lattice, args.cluster_switch = "table_ucg_bethe_density", "fcc", True
    if args.ncell is None:  # an explicit --ncell scales a preset down (tests, rehearsals)
        args.ncell = preset_ncell
    return args


# ------------------------------------------------------------------------------------------ launching N > 1 ranks

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` with no rendezvous environment: start the N ranks as fresh child processes.  This
    parent never touches the GPU (no HIP call, no torch import); it relays rank 0's JSON line and fails if any rank fails."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    failed = None
    line = b""
    alive = list(range(n))
    while alive and failed is None:
        for r in list(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.remove(r)
            if rc != 0:
                failed = (r, rc)
        if alive and failed is None:
            time.sleep(0.2)
    if failed is not None:  # a dead rank leaves the others waiting in a collective: stop exactly the ranks started here
        for r in alive:
            procs[r].terminate()
        for r in alive:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        raise SystemExit(f"bench.py: rank {failed[0]} exited with code {failed[1]}")
    line = procs[0].stdout.read()
    sys.stdout.write(line.decode())
    sys.stdout.flush()


# --------------------------------------------------------------------------------------------------- CPU baseline

def cpu_baseline(pkg, deck, ncell, nsteps, dt, integrator="wall"):
    """oracle, reference order (sequential half list + scatter + reverse sum), 1 core"""
    orc = entry.load_oracle()
    beads = pkg.synth.make_beads(ncell, seed=12345)
    op = orc.Pair("table_ucgld")
    op.settings(deck.pair_style_args())
    op.coeff(deck.pair_coeff_args())
    op.init(2, 1.0, 1.0)
    sim = orc.Sim(beads)
    sim.set_run_params(dt=dt, every=10, delay=0, check=1, mode=0)
    sim.attach(op, langevin=(1.0, 1.0, 1.0, 48279), nve="wall" if integrator == "wall" else True, ucgstate="ld")
    sim.setup(nsteps)
    t0 = time.perf_counter()
    sim.run(nsteps, 0)
    t = time.perf_counter() - t0
    info = sim.info()
    atom_steps = beads.n * nsteps / t
    return dict(seconds=t, n=beads.n, steps=nsteps, atom_steps_per_s=atom_steps, nhalf=info["nhalf"],
                ns_per_entry=t / nsteps / max(info["nhalf"], 1) * 1e9)


def cpu_baseline_threads(pkg, deck, ncell, nsteps, dt, nthreads, integrator="wall"):
    """P independent copies of the same scalar loop, one per host thread (the oracle is a C library:
    ctypes releases the GIL), each on its own periodic ncell^3 box with its own periodic-image ghosts and
    reverse sum -- the per-rank work of `mpirun -np P` of the reference, without the messages between the
    ranks (so an upper bound for it)."""
    import threading

    orc = entry.load_oracle()
    sims = []
    for t in range(nthreads):  # set-up is serial (the parsers use strtok, as the reference's do)
        beads = pkg.synth.make_beads(ncell, seed=12345 + t)
        op = orc.Pair("table_ucgld")
        op.settings(deck.pair_style_args())
        op.coeff(deck.pair_coeff_args())
        op.init(2, 1.0, 1.0)
        sim = orc.Sim(beads)
        sim.set_run_params(dt=dt, every=10, delay=0, check=1, mode=0)
        sim.attach(op, langevin=(1.0, 1.0, 1.0, 48279 + t), nve="wall" if integrator == "wall" else True, ucgstate="ld")
        sim.setup(nsteps)
        sims.append((sim, op, beads.n))
    threads = [threading.Thread(target=lambda s=s: s[0].run(nsteps, 0)) for s in sims]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    t = time.perf_counter() - t0
    n = sum(s[2] for s in sims)
    return dict(seconds=t, n=n, steps=nsteps, atom_steps_per_s=n * nsteps / t, threads=nthreads, per_thread=sims[0][2])


def host_cores():
    """(physical cores, logical CPUs, model) of the box from lscpu; None where it cannot be read"""
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
    except Exception:  # noqa: BLE001
        return None, os.cpu_count(), None
    kv = {}
    for ln in txt.splitlines():
        if ":" in ln:
            k, v = ln.split(":", 1)
            kv[k.strip()] = v.strip()
    try:
        phys = int(kv["Socket(s)"]) * int(kv["Core(s) per socket"])
    except Exception:  # noqa: BLE001
        phys = None
    return phys, os.cpu_count(), kv.get("Model name")


# ---------------------------------------------------------------------------------------------- problem definition

def make_problem(args, pkg, workdir):
    """deck + beads (+ the files of fix cluster_switch) of the selected workload"""
    synth = pkg.synth
    cs = None
    if args.cluster_switch:
        dens = (11.3, 1.5) if args.style == "table_ucg_bethe_density" else None
        kw = ("method", "bethe", "pseudo", "yes", "prior", "ucgl") if args.style == "table_ucg_bethe" else ()
        deck = synth.make_multi_deck(workdir, 2, args.tabstyle, args.tablength, density=dens,
                                     extra11=0.05 if dens else 0.0, extra_keywords=kw, n_file=2000)
        beads = synth.make_beads(args.ncell, seed=12345, lattice=args.lattice)
        msz = CLUSTER_SWITCH["molecule_size"]
        beads.ntypes = 4
        beads.mass = np.array([0.0, 1.0, 1.0, 1.0, 1.0])
        beads.molecule = ((beads.tag - 1) // msz + 1).astype(np.int32)
        rng = np.random.default_rng(777)
        mtype = rng.integers(1, 3, size=int(beads.molecule.max()) + 1)  # one type per molecule: wholly ON or OFF
        beads.type = mtype[beads.molecule].astype(np.int32)
        rates, contacts = synth.write_cluster_switch_files(workdir, CLUSTER_SWITCH["prob_on"], [1], [2], [(1, 1)])
        mol_seed = int(beads.molecule[np.flatnonzero(beads.type == 1)[0]])
        cs = dict(mol_seed=mol_seed, rates=rates, contacts=contacts, **CLUSTER_SWITCH)
    else:
        if args.style == "table_ucg_bethe":
            deck = synth.make_deck(workdir, args.tabstyle, args.tablength,
                                   extra_keywords=("method", "bethe", "pseudo", "yes", "prior", "ucgl"))
        elif args.style == "table_ucg_bethe_density":
            deck = synth.make_deck(workdir, args.tabstyle, args.tablength, density=(11.3, 1.5), extra11=0.05)
        else:
            deck = synth.make_deck(workdir, args.tabstyle, args.tablength)
        beads = synth.make_beads(args.ncell, seed=12345, lattice=args.lattice)
    return deck, beads, cs


def make_pair(capi, ctx, args, deck):
    pair = capi.Pair(ctx, args.style)
    pair.settings(deck.pair_style_args())
    if hasattr(deck, "pair_coeff_commands"):
        for cmd in deck.pair_coeff_commands():
            pair.coeff(cmd, deck.ntypes)
        pair.init(deck.ntypes, 1.0)
    else:
        pair.coeff(deck.pair_coeff_args())
        pair.init(2, 1.0)
    return pair


def attach_fixes(ctx, args, rank=0):
    """the fixes of the workload in deck order; returns (use_langevin, use_ucgstate)"""
    if args.style == "table_ucgld":
        ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279, me=rank)
        ctx.fix_ucgstate("ld", me=rank)
    elif args.style == "table_ucg_bethe":
        ctx.fix_ucgstate(None, me=rank)  # state = round(ucgp), ucgl = ucgp (the prior of the next step)
    else:
        ctx.fix_ucgstate("mc", 9127, 0.01, me=rank)
    if args.integrator == "wall":
        ctx.fix_nve_ucgld_wall_hard(False, 0.1)
    return args.style == "table_ucgld", True


def apply_env_options(ctx):
    # lanes per bead: 0 = chosen from the bead count (1 at 1 M beads; more for boxes too small to fill 256 CUs)
    ctx.set_option("gather_slots", int(os.environ.get("UCG_GATHER_SLOTS", "0")))
    if os.environ.get("UCG_FMA_CONTRACT"):  # NOT the bit-exact path: see DESIGN.md 4.1; never the default
        ctx.set_option("fma_contract", int(os.environ["UCG_FMA_CONTRACT"]))
    for env, opt in (("UCG_POST_IN_PAIR", "post_in_pair"), ("UCG_STAGE_OWN", "stage_own"), ("UCG_PAIR_VROW", "pair_vrow"),
                     ("UCG_HOT_BLOCK", "hot_block"), ("UCG_KIND_BLOCKS", "kind_blocks"), ("UCG_STREAM_ROWS", "stream_rows"), ("UCG_GENERIC_KERNELS", "generic_kernels"),
                     ("UCG_ROWS_SORT_R2", "rows_sort_r2")):  # (the last one: an experiment, not the specification's row order)
        if os.environ.get(env):
            ctx.set_option(opt, int(os.environ[env]))


def pinned_mirrors(n):
    """the caller's arrays of the owned atoms (LAMMPS' atom->x ...), page-locked so that the mirror copies run at PCIe speed"""
    import ctypes as C

    M = dict(x=np.zeros((n, 3)), v=np.zeros((n, 3)), f=np.zeros((n, 3)), ucgstate=np.zeros(n, np.int32),
             num_ucgstates=np.zeros(n, np.int32), ucgl=np.zeros(n), ucgvl=np.zeros(n), ucgp=np.zeros(n), ucgforce=np.zeros(n),
             scores=np.zeros((n, 2)))
    pinned = 0
    try:
        hip = C.CDLL("libamdhip64.so")
        hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
        for a in M.values():
            pinned += hip.hipHostRegister(a.ctypes.data, a.nbytes, 0) == 0
    except OSError:
        hip = None

    def release():
        if hip is not None:
            for a in M.values():
                hip.hipHostUnregister(C.c_void_p(a.ctypes.data))
    return M, pinned == len(M), release


def dropin_leg(ctx, pair, n, steps, integrator, use_lang, use_st, resident_ms_per_step):
    """The SAME workload, continued on the same context, as a LAMMPS run would drive the package: one C-ABI call per hook in
    upstream Verlet's order (ucg_verlet_hooks_run, csrc/ucg_host.hip), device arrays authoritative between the hooks, the
    caller's arrays bound as host mirrors and synchronised before every re-neighbouring (what exchange / borders read) and
    every 100 steps (a thermo / dump step)."""
    M, pinned, release = pinned_mirrors(n)
    ctx.host_bind(M)
    kw = dict(nve="wall" if integrator == "wall" else True, langevin=use_lang, ucgstate=use_st)
    legs = {}
    for name, on_re in (("host_reneighbour", True), ("device_reneighbour", False)):
        ctx.synchronize()
        t0 = time.perf_counter()
        st = ctx.verlet_hooks_run(pair, steps, sync_every=100, sync_on_reneighbour=on_re, **kw)
        ctx.synchronize()
        el = time.perf_counter() - t0
        legs[name] = dict(value=steps / el, ms_per_step=1e3 * el / steps, vs_resident=resident_ms_per_step / (1e3 * el / steps),
                          rebuilds=st["rebuilds"], host_syncs=st["syncs"], downloads=st["downloads"], uploads=st["uploads"])
    t2 = time.perf_counter()
    ctx.host_sync(type(ctx).F_ALL)
    t3 = time.perf_counter()
    ctx.host_bind(None)
    release()
    a = legs["host_reneighbour"]
    return {"value": a["value"], "unit": "timesteps/s", "steps": steps, "ms_per_step": a["ms_per_step"], "vs_resident": a["vs_resident"],
            "rebuilds": a["rebuilds"], "host_syncs": a["host_syncs"], "downloads": a["downloads"], "uploads": a["uploads"],
            "pinned_host_arrays": bool(pinned), "one_full_sync_ms": 1e3 * (t3 - t2),
            "device_reneighbour": legs["device_reneighbour"],
            "note": "hook by hook in upstream Verlet's order through the C ABI (initial_integrate, re-neighbour decision on the "
                    "device, re-neighbouring or forward halo, Pair::compute, post_force hooks, final_integrate: one call and at "
                    "least one kernel each, no epilogue fusion); the caller's pinned arrays are bound as host mirrors "
                    "(ucg_host_bind): nothing crosses PCIe on an ordinary step.  `value`: mirrors synchronised before every "
                    "re-neighbouring (x v ucgstate ucgl ucgvl ucgp, what LAMMPS' host-side exchange / borders read) and completely "
                    "every 100 steps (thermo / dump); `device_reneighbour`: only the latter (the package re-neighbours on its own). "
                    "The re-neighbouring itself runs on the device in both.  A reported leg: the headline `value` is the resident loop"}


def run_single(args, pkg, capi, deck, beads, cs, local_rank, steps, warmup, integrator, dropin_steps=0):
    """the resident single-GPU loop (ucg_md_run); returns the measurement dict"""
    import torch

    ctx = capi.Context(local_rank, dt=0.002)
    apply_env_options(ctx)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
    pair = make_pair(capi, ctx, args, deck)
    a2 = argparse.Namespace(**vars(args))
    a2.integrator = integrator
    use_lang, use_st = attach_fixes(ctx, a2)
    if cs:
        ctx.fix_cluster_switch(cs["mol_seed"], 0, cs["cutoff"], cs["seed"], cs["switch_freq"], cs["rates"], cs["contacts"])
    ctx.md_attach(pair, nve="wall" if integrator == "wall" else True, langevin=use_lang, ucgstate=use_st)
    equil = getattr(args, "equilibrate", 0)
    ctx.md_setup(equil + warmup + steps)
    ctx.md_run(equil + warmup, 0)  # input preparation (melt the lattice), then the warm-up
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    info0 = ctx.md_info()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.md_run(steps, 0)
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    launches, pair_ms = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    pair.check_errors()
    info = ctx.md_info()
    out = dict(elapsed=elapsed, n=beads.n, pair_launches=launches, pair_ms=pair_ms, list_entries=info["list_entries"],
               nghost=info["nghost"], rebuilds=info["nrebuild"] - info0["nrebuild"], maxrow=info["maxrow"],
               virtual_rows=bool(pair.sum_fixed), lanes_per_bead=pair.gather_slots)
    if cs:
        out["cluster_switch_vector"] = [float(v) for v in ctx.fix_cluster_switch_vector()]
    if dropin_steps > 0 and not cs:
        out["dropin"] = dropin_leg(ctx, pair, beads.n, dropin_steps, integrator, use_lang, use_st, 1e3 * elapsed / steps)
    pair.close()
    ctx.close()
    return out


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        spawn_ranks(args.gpus)  # before anything here touches the GPU
        return

    # exactly ONE line on stdout (the JSON): everything the libraries print there (RCCL / gloo banners) is sent
    # to stderr while the benchmark runs
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch

    ndev = torch.cuda.device_count()  # does not initialise the GPU
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the UCG hot path has no CPU fallback")
    # fewer GPUs than ranks (a one-GPU box rehearsing the N > 1 path): the ranks share the GPUs and the halo is
    # host-staged over gloo -- RCCL cannot put two ranks on one device.  Same code path otherwise.
    shared = world > ndev
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    dist = None
    force_multi = os.environ.get("UCG_FORCE_MULTI") == "1"  # exercise the decomposed path (RCCL transport) with 1 rank
    if world > 1 or force_multi:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
        # torch.distributed (gloo) only carries the rendezvous, the barriers around the timed region and rank 0's
        # RCCL id; the halo itself is moved by the library (RCCL called from its C++ step loop), or -- ranks sharing
        # a GPU -- by host-staged gloo callbacks
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = entry.load_package()
    capi = pkg.capi
    dt = 0.002
    workdir = tempfile.mkdtemp(prefix=f"ucgbench_r{rank}_")
    import atexit
    import shutil
    atexit.register(shutil.rmtree, workdir, True)  # the generated table / settings files
    deck, beads, cs = make_problem(args, pkg, workdir)

    nve_leg = None
    if world > 1 or force_multi:
        from lammps_ucg_dev_amd import multi  # spatial decomposition + RCCL halo

        result = multi.run_bench(args, deck, beads, cs, rank, world, device_index, dist, shared,
                                 make_pair=lambda ctx: make_pair(capi, ctx, args, deck),
                                 attach_fixes=lambda ctx: attach_fixes(ctx, args, rank),
                                 apply_options=apply_env_options)
    else:
        result = run_single(args, pkg, capi, deck, beads, cs, device_index, args.steps, args.warmup, args.integrator,
                            dropin_steps=args.dropin_steps)
        if args.integrator == "wall" and not args.no_nve_leg and args.style == "table_ucgld" and not cs:
            # the integrator north_star names, on the same beads: a SHORT leg (it is not stationary, see --integrator)
            k = max(50, min(args.steps, 300))
            r2 = run_single(args, pkg, capi, deck, beads, None, device_index, k, min(args.warmup, 100), "nve")
            nve_leg = {"integrator": "fix nve/ucgld (UCG/fix_nve_ucgld.cpp:44-153)", "value": k / r2["elapsed"],
                       "unit": "timesteps/s", "steps": k, "warmup": min(args.warmup, 100),
                       "pair_avg_launch_us": r2["pair_ms"] / max(r2["pair_launches"], 1) * 1e3,
                       "note": "short by design: lambda is unbounded under this integrator and the melt leaves the tables' "
                               "range after ~1000 steps (DESIGN.md section 6)"}

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    n = result["n"]
    steps_per_s = args.steps / result["elapsed"]
    # the roofline line is about ONE launch of the pair kernel: rank 0's share of the beads for N > 1
    e_half = result.get("rank0_list_entries", result["list_entries"]) / 2.0
    n_launch = result.get("rank0_nlocal", n)
    pair_avg_s = (result["pair_ms"] / max(result["pair_launches"], 1)) * 1e-3
    if args.style == "table_ucg_bethe_density":
        alg_bytes = 116.0 * 2.0 * e_half + 220.0 * n_launch  # 116 E_full + 220 N (three passes)
        alg_note = "SURVEY.md 8(d): 116 B x full-list entries + 220 B x beads (three passes)"
    else:
        alg_bytes = 44.0 * e_half + 96.0 * n_launch  # B_alg = 44 E + 96 N (E = half-list entries)
        alg_note = "SURVEY.md 8(d): 44 B x half-list entries + 96 B x beads"
    achieved = alg_bytes / pair_avg_s / 1e9 if pair_avg_s > 0 else 0.0
    integ = "nve/ucgld/wall/hard" if args.integrator == "wall" else "nve/ucgld"
    style_txt = {"table_ucgld": "table_ucgld + INTEG + ucgld/langevin + ucgstate ld",
                 "table_ucg_bethe": "table_ucg_bethe method bethe pseudo yes prior ucgl + INTEG + ucgstate",
                 "table_ucg_bethe_density": "table_ucg_bethe_density + INTEG + ucgstate mc 9127 0.01"}[args.style]
    style_txt = style_txt.replace("INTEG", integ) + (" + cluster_switch" if cs else "")
    wl_fix = {"table_ucgld": "fix INTEG + fix ucgld/langevin 1.0 1.0 1.0 48279 + fix ucgstate ld; ",
              "table_ucg_bethe": "method bethe pseudo yes prior ucgl + fix INTEG + fix ucgstate; ",
              "table_ucg_bethe_density": "density 11.3 1.5 + fix INTEG + fix ucgstate mc 9127 0.01; "}[args.style].replace("INTEG", integ)
    if cs:
        wl_fix += (f"2 actual atom types (ON / OFF), molecules of {cs['molecule_size']} beads, fix cluster_switch {cs['mol_seed']} 0 "
                   f"{cs['cutoff']} {cs['seed']} {cs['switch_freq']} (probON {cs['prob_on']}); ")
    if world > 1 or "grid" in result:  # (UCG_FORCE_MULTI=1: the decomposed path with one rank)
        par = (f"spatial decomposition {'x'.join(map(str, result['grid']))} bricks, one process per GPU, forward halo = one "
               "neighbour all-to-all per step, no reverse halo; transport: " + result.get("transport", "RCCL"))
    else:
        par = "1 GPU"
    out = {
        "metric": "timesteps/sec at " + ("1M" if n == 1000000 else str(n)) + " UCG beads (" + style_txt + ")",
        "value": steps_per_s,
        "unit": "timesteps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * result["elapsed"] / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "atom_steps_per_s": steps_per_s * n,
        "config": {
            "workload": f"{n} beads ({args.lattice} lattice {args.ncell}^3 + jitter), rho*=0.8, rc=2.5, skin=0.3, dt=0.002, "
                        f"pair_style {args.style} {args.tabstyle} {args.tablength} (2-state, LJ-like tables) + " + wl_fix +
                        f"neigh_modify every 10 check yes; input = the lattice after {args.equilibrate} steps of this same path "
                        f"(SURVEY.md 8d); rebuilds inside the timed region: {result['rebuilds']}",
            "beads": n,
            "full_list_entries": int(result["list_entries"]),
            "ghosts": int(result["nghost"]),
            "parallelism": par,
        },
        # N > 1 (and UCG_FORCE_MULTI=1): what moved the halo, asked of RCCL itself -- a host-staged rehearsal says false / 0
        **({"rccl": bool(result["rccl"]), "rccl_nranks": int(result["rccl_nranks"])} if "rccl" in result else {}),
        "roofline": {
            "bound": "hbm",
            "kernel": ("k_density_pass1+2+3" if args.style == "table_ucg_bethe_density"
                       else (f"k_pair_vrow<{args.style}>" if result.get("virtual_rows") else f"k_pair_gather<{args.style}>")
                       + " (with the fused per-bead epilogue)"),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes": alg_note,
            "compulsory_bytes_per_launch": 4.0 * e_half + 96.0 * n_launch,  # B_min = 4 E + 96 N
            "avg_launch_us": pair_avg_s * 1e6,
            "launches": int(result["pair_launches"]),
        },
    }
    if args.style != "table_ucg_bethe_density":
        # the launch also runs the per-bead hooks in its epilogue: f / ucgforce / scores (48 B per
        # bead) stay in registers and the hooks move 124 B per bead of their own; side figure, never `frac`
        b172 = 44.0 * e_half + (96.0 - 48.0 + 124.0) * n_launch
        out["roofline"]["with_fused_hooks"] = {
            "algorithmic_bytes_per_launch": b172, "frac": b172 / pair_avg_s / 1e9 / HBM_PEAK_GBS if pair_avg_s > 0 else 0.0,
            "note": "44 E + 172 N: the pair loop's bytes minus the 48 B kept in registers plus the 124 B of the fused hooks"}
    if "virtual_rows" in result:
        out["config"]["pair_kernel"] = {"virtual_rows": bool(result["virtual_rows"]), "lanes_per_bead": int(result["lanes_per_bead"]),
                                        "note": "virtual_rows: the pairs of two beads of one 512-bead workgroup block are evaluated once, "
                                                "on balanced virtual rows with order-free fixed sums (ucg_pair_vrow.hip, DESIGN.md 4.1); "
                                                "false: full-row gather kernel, lanes_per_bead lanes per bead (option pair_vrow 0)"}
    if cs and "cluster_switch_vector" in result:
        out["config"]["cluster_switch_vector"] = result["cluster_switch_vector"]
    if "small_messages" in result:
        out["config"]["small_messages"] = result["small_messages"]  # counts / flags of re-neighbouring steps: route taken
    # HBM traffic and VALU instruction counts of the pair kernel: measured in separate rocprofv3 --pmc passes of this
    # same command (tools/profile_pmc.sh) and committed under profiles/ with the commit and kernel they were taken on;
    # used only when that stamp names this workload
    pmc = load_pmc_stamp(args, world, cs)
    if pmc:
        out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
        out["roofline"]["traffic_note"] = pmc["note"]
        if pmc.get("valu_insts_per_launch"):
            v = pmc["valu_insts_per_launch"]
            t_valu = v * 4.0 / SIMDS / (CLOCK_GHZ * 1e9)
            out["roofline"]["valu"] = {
                "bound": "fp64 VALU issue", "wave_instructions_per_launch": v, "min_us": t_valu * 1e6,
                "frac": t_valu / pair_avg_s if pair_avg_s > 0 else 0.0,
                "frac_at_measured_clock": t_valu * (CLOCK_GHZ / MEASURED_CLOCK_GHZ) / pair_avg_s if pair_avg_s > 0 else 0.0,
                "note": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz) over the measured launch time: the on-chip bound that "
                        "binds this kernel together with the LDS (HBM traffic is a third of the algorithmic bytes); under this "
                        f"kernel the chip holds {MEASURED_CLOCK_GHZ} GHz (GRBM_GUI_ACTIVE, tools/profile_clock.sh), hence the second figure"}
    if nve_leg:
        out["integrator_nve"] = nve_leg
    if "dropin" in result:
        out["dropin"] = result["dropin"]
    if not args.no_cpu_baseline and args.style == "table_ucgld" and world == 1 and not cs:  # rank 0 at N = 1 only
        sdeck = deck
        cb = cpu_baseline(pkg, sdeck, args.cpu_ncell, args.cpu_steps, dt, args.integrator)
        phys, logical, model = host_cores()
        one = {
            "value": cb["atom_steps_per_s"] / n,
            "unit": "timesteps/s",
            "cores": 1,
            "kind": "port",
            "sample": f"oracle reference-order loop (half list, scalar, gcc -O2 -ffp-contract=off), {cb['n']} beads x "
                      f"{cb['steps']} full steps in {cb['seconds']:.2f} s = {cb['atom_steps_per_s']:.4g} bead-steps/s "
                      f"({cb['ns_per_entry']:.1f} ns per half-list entry per step), scaled to {n} beads",
        }
        nthreads = args.cpu_threads or min(os.cpu_count() or 1, 16)
        if nthreads > 1:
            # the reference runs one MPI rank per core: P copies of the same scalar loop, one per thread
            mt = cpu_baseline_threads(pkg, sdeck, 40, 2 * args.cpu_steps, dt, nthreads, args.integrator)
            out["cpu_baseline"] = {
                "value": mt["atom_steps_per_s"] / n,
                "unit": "timesteps/s",
                "cores": nthreads,
                "kind": "port",
                "sample": f"{nthreads} host threads, each the oracle's reference-order loop on its own periodic box of "
                          f"{mt['per_thread']} beads (with its periodic-image ghosts and reverse sum) x {mt['steps']} full "
                          f"steps, no messages between them: an upper bound for mpirun -np {nthreads} of the reference, "
                          f"{mt['seconds']:.2f} s = {mt['atom_steps_per_s']:.4g} bead-steps/s, scaled to {n} beads",
                "one_core": one,
            }
        else:
            out["cpu_baseline"] = one
        cbl = out["cpu_baseline"]
        cbl["host"] = {"physical_cores": phys, "logical_cpus": logical, "model": model,
                       "note": "lscpu of this box; a one-GPU box gives a job 16 of them (the pool's CPU share per GPU), "
                               "which is what `cores` uses"}
        cbl["gpu_over_cpu"] = {"vs_1_core": out["value"] / one["value"], f"vs_{cbl['cores']}_cores": out["value"] / cbl["value"]}
    emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def load_pmc_stamp(args, world, cs):
    """profiles/rNN_pmc_<style>.json (the newest round): {"commit", "kernel", "workload", "traffic_bytes_per_launch", "valu_insts_per_launch", "note"}"""
    if world != 1 or cs or args.tabstyle != "spline" or args.tablength != 1024 or args.lattice != "sc" or args.ncell != 100:
        return None
    import glob
    stamps = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_{args.style}.json")))  # named per round: the newest
    if not stamps:
        return None
    path = stamps[-1]
    with open(path) as fh:
        d = json.load(fh)
    if d.get("workload") != f"{args.style} {args.tabstyle} {args.tablength} sc {args.ncell}":
        return None
    d["note"] = (f"rocprofv3 PMC of kernel {d.get('kernel')} at commit {d.get('commit')} (profiles/{os.path.basename(path)}): "
                 "FETCH_SIZE x2 per the gfx950 calibration + WRITE_SIZE; SQ_INSTS_VALU")
    return d


if __name__ == "__main__":
    main()
