#!/usr/bin/env python3
"""Drop-in boundary (LAMMPS owns the host arrays): per force evaluation the glue uploads the comm
fields of owned + ghost beads, runs the pair kernel and downloads f / ucgforce / scores.
Measures that PCIe-inclusive rate at 1 M beads (never bench.py's `value`)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
capi, synth = pkg.capi, pkg.synth
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 100
beads = synth.make_beads(ncell, seed=12345)
deck = synth.make_deck(tempfile.mkdtemp(), "spline", 1024)
ctx = capi.Context(0, dt=0.002)
ctx.upload_beads(beads)
ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
ctx.neigh_rebuild()
pair = capi.Pair(ctx, "table_ucgld")
pair.settings(deck.pair_style_args()); pair.coeff(deck.pair_coeff_args()); pair.init(2, 1.0)
A = ctx.atoms_download(with_ghosts=True)
x, st, l, p = A["x"].copy(), A["ucgstate"].copy(), A["ucgl"].copy(), A["ucgp"].copy()
nall = len(x)
for rep in range(2):
    ctx.atoms_upload_comm(x, st, l, p); pair.compute(0, 0); ctx.atoms_download()
ctx.synchronize()
N = 10
t0 = time.perf_counter()
for rep in range(N):
    ctx.atoms_upload_comm(x, st, l, p)
    pair.compute(0, 0)
    out = ctx.atoms_download()
ctx.synchronize()
dt = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for rep in range(N):
    pair.compute(0, 0)
ctx.synchronize()
dk = (time.perf_counter() - t0) / N
print(f"beads {beads.n} (+{nall - beads.n} ghosts): drop-in force evaluation {dt * 1e3:.2f} ms "
      f"(upload 44 B x {nall} + kernel + download all owned fields), kernel alone {dk * 1e3:.3f} ms "
      f"-> {1.0 / dt:.1f} evaluations/s PCIe-inclusive vs {1.0 / dk:.1f} resident")
