"""A plain-C caller of include/ucg_hip.h (tests/c_abi/ucg_c_caller.c): compiled with gcc -std=c99 -Wall -Werror against
the header and linked with libucg_hip.so, so the PROTOTYPES are checked by a C compiler (ctypes only checks names).
CPU tier: it compiles and links.  GPU tier: it runs the golden decks through the ABI and finds the committed bits."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import test_golden as tg
import util

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "lammps-ucg-dev_amd")


def _build(outdir):
    exe = os.path.join(outdir, "ucg_c_caller")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(HERE, "c_abi", "ucg_c_caller.c"), "-o", exe, "-L", PKG, "-lucg_hip",
                           "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_caller_compiles_and_links_against_the_header():
    d = tempfile.mkdtemp(prefix="ucgc_")
    exe = _build(d)
    assert os.path.exists(exe)
    # every ucg_* symbol the C program references is resolved by the library at link time (no -shared trickery)
    out = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    used = sorted({ln.split()[-1].split("@")[0] for ln in out.splitlines() if " ucg_" in ln})
    assert len(used) >= 20 and "ucg_pair_compute" in used and "ucg_neigh_upload_full" in used
    assert {"ucg_host_bind", "ucg_host_sync", "ucg_host_status", "ucg_decide_local", "ucg_halo_forward"} <= set(used)


def _write_case(path, pkg, g, name):
    case = g["cases"][name]
    beads = tg._beads(pkg, g)
    deck = tg._deck(case)
    style = {"table_ucgld": 0, "table_ucg_bethe": 1, "table_ucg_bethe_density": 2}[case["style"]]
    with open(path, "wb") as fh:
        fh.write(f"UCGCASE1 {beads.n} {beads.ntypes} {style} 10\n".encode())
        for ln in (deck.table_file, deck.conf_file, deck.tabstyle, str(deck.tablength), " ".join(case["extra"])):
            fh.write((ln + "\n").encode())
        f64 = lambda a: fh.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())  # noqa: E731
        i32 = lambda a: fh.write(np.ascontiguousarray(a, dtype=np.int32).tobytes())    # noqa: E731
        for a in (beads.boxlo, beads.boxhi, beads.x, beads.v, beads.ucgl, beads.ucgvl, beads.ucgml, beads.ucgp, beads.mass):
            f64(a)
        for a in (beads.type, beads.tag, beads.mask, beads.ucgstate):
            i32(a)
        s0, s1 = case["setup"], case["after10"]
        i32(tg._i(s0["tag"]))
        for k in ("f", "scores", "ucgforce", "ucgp"):
            f64(tg._f(s0[k]))
        i32(tg._i(s1["tag"]))
        i32(tg._i(s1["ucgstate"]))
        for k in ("x", "v", "ucgl"):
            f64(tg._f(s1[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ucgld_spline1024", "bethe_pseudo_yes", "density"])
def test_c_caller_reproduces_golden_bits(pkg, name):
    d = tempfile.mkdtemp(prefix="ucgc_")
    exe = _build(d)
    case = os.path.join(d, name + ".case")
    _write_case(case, pkg, tg._load(), name)
    r = subprocess.run([exe, case], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatching groups" in r.stdout and "hook-by-hook run" in r.stdout


# ---- the decomposed step loop from a plain-C multi-process caller (tests/c_abi/ucg_c_world2.c)

def _build_world(outdir):
    exe = os.path.join(outdir, "ucg_c_world2")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(HERE, "c_abi", "ucg_c_world2.c"), "-o", exe, "-L", PKG, "-lucg_hip",
                           "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_world_caller_compiles_and_links_against_the_header():
    d = tempfile.mkdtemp(prefix="ucgcw_")
    exe = _build_world(d)
    out = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    used = {ln.split()[-1].split("@")[0] for ln in out.splitlines() if " ucg_" in ln}
    assert {"ucg_comm_attach_host", "ucg_comm_transport", "ucg_decomp_set", "ucg_md_run_until", "ucg_md_setup"} <= used


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_c_world_caller_finds_the_decomposed_oracles_bits(pkg, orc, world):
    """N processes (forked before any GPU call) drive the decomposed resident loop through the C
    ABI alone -- ucg_decomp_set + ucg_comm_attach_host + ucg_md_setup / ucg_md_run_until, the calls of the glue's
    `run_style verlet/ucg/gpu comm mpi` -- and every rank finds, in its local order, the bits of the oracle's decomposed
    run (orc_world): forces / ucgforce / scores at setup; tags, states, positions, velocities and lambda after a
    thermostatted run with migration (fix ucgld/langevin + fix ucgstate mc + fix nve/ucgld/wall/hard)."""
    nsteps = 240
    deck = util.make_deck("spline", 1024)
    beads = pkg.synth.make_beads(10, seed=5)
    grid = pkg.multi.choose_procgrid(world)
    op = util.oracle_pair("table_ucgld", deck)
    w = orc.World(beads, grid)
    w.set_run_params(dt=0.004, every=2, delay=0, check=1)
    w.attach(op, langevin=(1.0, 1.0, 1.0, 48279), nve="wall", ucgstate=("mc", 9127, 0.3))
    assert w.setup(nsteps) == 0
    exp0 = [w.rank_arrays(r) for r in range(world)]
    exp0 = [{k: np.array(v, copy=True) if isinstance(v, np.ndarray) else v for k, v in e.items()} for e in exp0]
    assert w.run(nsteps) == 0
    exp1 = [w.rank_arrays(r) for r in range(world)]
    assert any(e0["nlocal"] != e1["nlocal"] for e0, e1 in zip(exp0, exp1))   # beads did migrate between the bricks
    d = tempfile.mkdtemp(prefix="ucgcw_")
    exe = _build_world(d)
    case = os.path.join(d, "world.case")
    with open(case, "wb") as fh:
        fh.write(f"UCGWORLD1 {beads.n} {beads.ntypes} {nsteps} {grid[0]} {grid[1]} {grid[2]}\n".encode())
        for ln in (deck.table_file, deck.conf_file, deck.tabstyle, str(deck.tablength)):
            fh.write((ln + "\n").encode())
        f64 = lambda a: fh.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())  # noqa: E731
        i32 = lambda a: fh.write(np.ascontiguousarray(a, dtype=np.int32).tobytes())    # noqa: E731
        for a in (beads.boxlo, beads.boxhi, beads.x, beads.v, beads.ucgl, beads.ucgvl, beads.ucgml, beads.ucgp, beads.mass):
            f64(a)
        for a in (beads.type, beads.tag, beads.mask, beads.ucgstate):
            i32(a)
        for r in range(world):
            e0, e1 = exp0[r], exp1[r]
            n0, n1 = e0["nlocal"], e1["nlocal"]
            i32([n0]); i32(e0["tag"][:n0])
            f64(e0["f"][:n0]); f64(e0["ucgforce"][:n0]); f64(e0["scores"][:n0])
            i32([n1]); i32(e1["tag"][:n1]); i32(e1["ucgstate"][:n1])
            f64(e1["x"][:n1]); f64(e1["v"][:n1]); f64(e1["ucgl"][:n1])
    r = subprocess.run([exe, case, str(world)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"all {world} ranks found the decomposed oracle's bits" in r.stdout
