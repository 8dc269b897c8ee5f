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
