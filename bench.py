#!/usr/bin/env python3
"""bench.py -- UCG hot path on MI355X: timesteps/s at 1 M UCG beads.

  python bench.py --gpus N --steps K --warmup W

One "step" is one full Verlet step of the resident path on synthetic input:
fix nve/ucgld/wall/hard initial_integrate -> re-neighbour decision (every 10 steps, rebuild when a
bead moved skin/2) / halo refresh -> pair_style table_ucgld (spline 1024, 2-state, the
north-star neighbour loop) -> fix ucgld/langevin -> fix ucgstate ld -> final_integrate.
(The hard-wall variant of the lambda integrator: plain fix nve/ucgld lets lambda drift without bound and the
melt breaks down after ~1000 steps, in the reference as here -- see --integrator.)
Workload (BASELINE.md config 4 at N GPUs, config-2 styles at 1 M beads): 100^3 beads at
rho* = 0.8, rc = 2.5, skin = 0.3, dt = 0.002, fp64 throughout.  For N > 1 the SAME 1 M beads
are split across the ranks (strong scaling).  Inputs are resident in HBM before timing.

The JSON line also carries
  roofline     -- the pair kernel's algorithmic bytes (44 B per half-list entry + 96 B per bead,
                  SURVEY.md 8d) / its mean duration from HIP events on its own stream, vs 8 TB/s
  cpu_baseline -- the oracle's reference-order (half list, scalar) loop timed on this box's host
                  cores on a bounded sample; a reported baseline, not a target.
"""
import argparse
import json
import os
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
    ctypes releases the GIL), each on its own periodic ncell^3 box -- what `mpirun -np P` of the
    reference does per rank, without the halo exchange (so an upper bound for it)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps (SURVEY.md 8d: >= 1000, re-neighbouring included)")
    ap.add_argument("--warmup", type=int, default=200, help="untimed steps first (SURVEY.md 8d: 200 warm-up steps on the path itself)")
    ap.add_argument("--ncell", type=int, default=100, help="beads = ncell^3 (default 100 -> 1 M)")
    ap.add_argument("--tabstyle", default="spline")
    ap.add_argument("--tablength", type=int, default=1024)
    ap.add_argument("--style", default="table_ucgld",
                    choices=["table_ucgld", "table_ucg_bethe", "table_ucg_bethe_density"],
                    help="pair style of the 1-GPU leg (default: the headline table_ucgld workload; the others are "
                         "BASELINE.md configs 3 and 5 at 1 M beads, reported with their own algorithmic bytes)")
    ap.add_argument("--lattice", default="sc", choices=["sc", "fcc"],
                    help="sc: ncell^3 beads (default); fcc: 4 ncell^3 beads (BASELINE.json's 4 M-bead configuration = --ncell 100 --lattice fcc)")
    ap.add_argument("--integrator", default="wall", choices=["wall", "nve"],
                    help="wall = fix nve/ucgld/wall/hard (default), nve = fix nve/ucgld.  The latter never clamps lambda "
                         "(UCG/fix_nve_ucgld.cpp:44-153): with the linear-in-lambda mixing nothing confines it, and after "
                         "~1000 steps of this melt it has drifted far enough outside [0, 1] for the mixed potentials to "
                         "turn attractive at contact -- the reference stops there with 'Pair distance < table inner "
                         "cutoff' (tools/stability.py).  The hard-wall variant of the same integrator is stable.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ncell", type=int, default=64)
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the multi-core baseline (0 = min(cores, 16))")
    args = ap.parse_args()

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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the UCG hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    force_multi = os.environ.get("UCG_FORCE_MULTI") == "1"  # exercise the decomposed path (RCCL transport) with 1 rank
    if world > 1 or force_multi:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = entry.load_package()
    capi, synth = pkg.capi, pkg.synth
    dt = 0.002
    workdir = tempfile.mkdtemp(prefix=f"ucgbench_r{rank}_")
    import atexit
    import shutil
    atexit.register(shutil.rmtree, workdir, True)  # the generated table / settings files
    if args.style == "table_ucg_bethe":
        deck = synth.make_deck(workdir, args.tabstyle, args.tablength,
                               extra_keywords=("method", "bethe", "pseudo", "yes", "prior", "ucgl"))
    elif args.style == "table_ucg_bethe_density":
        deck = synth.make_deck(workdir, args.tabstyle, args.tablength, density=(11.3, 1.5), extra11=0.05)
    else:
        deck = synth.make_deck(workdir, args.tabstyle, args.tablength)
    if args.style != "table_ucgld" and (world > 1 or force_multi):
        raise SystemExit("--style other than table_ucgld is a 1-GPU leg")

    if world > 1 or force_multi:
        from lammps_ucg_dev_amd import multi  # spatial decomposition + RCCL halo

        result = multi.run_bench(args, deck, rank, world, local_rank, dist)
    else:
        beads = synth.make_beads(args.ncell, seed=12345, lattice=args.lattice)
        ctx = capi.Context(local_rank, dt=dt)
        # lanes per bead: 0 = chosen from the bead count (1 at 1 M beads; more for boxes too small to fill 256 CUs)
        ctx.set_option("gather_slots", int(os.environ.get("UCG_GATHER_SLOTS", "0")))
        if os.environ.get("UCG_FMA_CONTRACT"):  # NOT the bit-exact path: see DESIGN.md 4.1; never the default
            ctx.set_option("fma_contract", int(os.environ["UCG_FMA_CONTRACT"]))
        if os.environ.get("UCG_POST_IN_PAIR"):
            ctx.set_option("post_in_pair", int(os.environ["UCG_POST_IN_PAIR"]))
        if os.environ.get("UCG_STAGE_OWN"):
            ctx.set_option("stage_own", int(os.environ["UCG_STAGE_OWN"]))
        ctx.upload_beads(beads)
        ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
        pair = capi.Pair(ctx, args.style)
        pair.settings(deck.pair_style_args())
        pair.coeff(deck.pair_coeff_args())
        pair.init(2, 1.0)
        if args.style == "table_ucgld":
            ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279)
            ctx.fix_ucgstate("ld")
        elif args.style == "table_ucg_bethe":
            ctx.fix_ucgstate(None)  # state = round(ucgp), ucgl = ucgp (the prior of the next step)
        else:
            ctx.fix_ucgstate("mc", 9127, 0.01)
        wall = args.integrator == "wall"
        if wall:
            ctx.fix_nve_ucgld_wall_hard(False, 0.1)
        ctx.md_attach(pair, nve="wall" if wall else True, langevin=args.style == "table_ucgld", ucgstate=True)
        ctx.md_setup(args.warmup + args.steps)
        ctx.md_run(args.warmup, 0)
        ctx.synchronize()
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        info0 = ctx.md_info()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.md_run(args.steps, 0)
        ctx.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        launches, pair_ms = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        pair.check_errors()
        info = ctx.md_info()
        result = dict(elapsed=elapsed, n=beads.n, pair_launches=launches, pair_ms=pair_ms,
                      list_entries=info["list_entries"], nghost=info["nghost"],
                      rebuilds=info["nrebuild"] - info0["nrebuild"], maxrow=info["maxrow"])

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
    if args.style == "table_ucg_bethe_density":
        alg_bytes = 116.0 * 2.0 * e_half + 220.0 * n_launch  # 116 E_full + 220 N (three passes)
        alg_note = "116 B x full-list entries + 220 B x beads (three passes)"
    else:
        # B_alg = 44 E + 96 N for the pair loop (E = half-list entries).  The launches of the
        # resident loop also run the per-bead hooks in their epilogue: f / ucgforce / scores (48 B
        # per bead) are then consumed in registers instead of being written, and the hooks add what they must move
        # themselves: read v 32 + ucgml 8 + mask 4 + draw 4, write v 32 + next x 32 + state 4 + ucgp 8 = 124 B.
        alg_bytes = 44.0 * e_half + (96.0 - 48.0 + 124.0) * n_launch
        alg_note = ("44 B x half-list entries + 172 B x beads: the pair loop's 44 E + 96 N minus the 48 B of f / ucgforce / "
                    "scores kept in registers plus the 124 B the fused per-bead hooks move")
    pair_avg_s = (result["pair_ms"] / max(result["pair_launches"], 1)) * 1e-3
    achieved = alg_bytes / pair_avg_s / 1e9 if pair_avg_s > 0 else 0.0
    out = {
        "metric": "timesteps/sec at " + ("1M" if n == 1000000 else str(n)) + " UCG beads (" + {
            "table_ucgld": "table_ucgld + INTEG + ucgld/langevin + ucgstate ld",
            "table_ucg_bethe": "table_ucg_bethe method bethe pseudo yes prior ucgl + INTEG + ucgstate",
            "table_ucg_bethe_density": "table_ucg_bethe_density + INTEG + ucgstate mc 9127 0.01"}[args.style].replace(
                "INTEG", "nve/ucgld/wall/hard" if args.integrator == "wall" else "nve/ucgld") + ")",
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
                        f"pair_style {args.style} {args.tabstyle} {args.tablength} (2-state, 4 LJ-like tables) + " + {
                            "table_ucgld": "fix INTEG + fix ucgld/langevin 1.0 1.0 1.0 48279 + fix ucgstate ld; ",
                            "table_ucg_bethe": "method bethe pseudo yes prior ucgl + fix INTEG + fix ucgstate; ",
                            "table_ucg_bethe_density": "density 11.3 1.5 + fix INTEG + fix ucgstate mc 9127 0.01; ",
                        }[args.style].replace("INTEG", "nve/ucgld/wall/hard" if args.integrator == "wall" else "nve/ucgld") +
                        "neigh_modify every 10 check yes; rebuilds inside the timed region: "
                        f"{result['rebuilds']}",
            "beads": n,
            "full_list_entries": int(result["list_entries"]),
            "ghosts": int(result["nghost"]),
            "parallelism": (f"spatial decomposition {'x'.join(map(str, result['grid']))} bricks, one process per GPU, "
                            "forward halo = one RCCL all_to_all per step, no reverse halo") if world > 1 else "1 GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("k_density_pass1+2+3" if args.style == "table_ucg_bethe_density"
                       else f"k_pair_gather<{args.style}> + fused per-bead epilogue"),
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes": alg_note,
            "pair_loop_only": {"algorithmic_bytes_per_launch": 44.0 * e_half + 96.0 * n_launch,
                               "note": "44 E + 96 N over the same launch time (which includes the fused hooks)",
                               "frac": (44.0 * e_half + 96.0 * n_launch) / pair_avg_s / 1e9 / HBM_PEAK_GBS if pair_avg_s > 0 else 0.0},
            "compulsory_bytes_per_launch": 4.0 * e_half + 96.0 * n_launch,  # B_min = 4 E + 96 N
            "avg_launch_us": pair_avg_s * 1e6,
            "launches": int(result["pair_launches"]),
        },
    }
    if "small_messages" in result:
        out["config"]["small_messages"] = result["small_messages"]  # counts / flags of re-neighbouring steps: route taken
    # HBM traffic of the pair kernel: measured in separate rocprofv3 --pmc passes of this same
    # command (tools/profile_pmc.sh) and committed under profiles/; valid for the default workload
    tname = {"table_ucgld": "r01_pair_traffic.json", "table_ucg_bethe": "r01_bethe_traffic.json",
             "table_ucg_bethe_density": "r01_density_traffic.json"}[args.style]
    tfile = os.path.join(ROOT, "profiles", tname)
    if (world == 1 and args.ncell == 100 and args.lattice == "sc" and args.tabstyle == "spline"
            and args.tablength == 1024 and os.path.exists(tfile)):
        with open(tfile) as fh:
            out["roofline"]["traffic"] = json.load(fh)["traffic_bytes_per_launch"]
        out["roofline"]["traffic_note"] = ("HBM bytes per launch from rocprofv3 PMC (FETCH_SIZE x2 per the gfx950 calibration + "
                                           "WRITE_SIZE), profiles/" + tname)
    if not args.no_cpu_baseline and args.style == "table_ucgld" and world == 1:  # rank 0 at N = 1 only
        cb = cpu_baseline(pkg, deck, args.cpu_ncell, args.cpu_steps, dt, args.integrator)
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
            mt = cpu_baseline_threads(pkg, deck, 40, 2 * args.cpu_steps, dt, nthreads, args.integrator)
            out["cpu_baseline"] = {
                "value": mt["atom_steps_per_s"] / n,
                "unit": "timesteps/s",
                "cores": nthreads,
                "kind": "port",
                "sample": f"{nthreads} host threads, each the oracle's reference-order loop on its own periodic box of "
                          f"{mt['per_thread']} beads x {mt['steps']} full steps (no halo exchange between them: an upper "
                          f"bound for mpirun -np {nthreads} of the reference), {mt['seconds']:.2f} s = "
                          f"{mt['atom_steps_per_s']:.4g} bead-steps/s, scaled to {n} beads",
                "one_core": one,
            }
        else:
            out["cpu_baseline"] = one
        cbl = out["cpu_baseline"]
        cbl["gpu_over_cpu"] = {"vs_1_core": out["value"] / one["value"], f"vs_{cbl['cores']}_cores": out["value"] / cbl["value"]}
    emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
