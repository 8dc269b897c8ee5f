#!/usr/bin/env python3
"""How much of the gather kernel's wavefront iterations are idle lanes?  For the bench workload after warm-up:
per 64-bead wavefront, max vs mean of (a) the row length and (b) the number of in-cutoff entries."""
import os
import sys
import tempfile

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
pair = capi.Pair(ctx, "table_ucgld")
pair.settings(deck.pair_style_args()); pair.coeff(deck.pair_coeff_args()); pair.init(2, 1.0)
ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279); ctx.fix_ucgstate("ld")
ctx.md_attach(pair, nve=True, langevin=True, ucgstate=True)
ctx.md_setup(100); ctx.md_run(55, 0)
il, nn, fi, ne = ctx.neigh_download()
A = ctx.atoms_download(with_ghosts=True)
x = A["x"]
n = len(nn)
rows = np.repeat(np.arange(n, dtype=np.int64), nn)
idx = (ne & 0x1FFFFFFF).astype(np.int64)
c0 = np.zeros(n, dtype=np.int64)
step = 8_000_000
for s in range(0, len(idx), step):
    d = x[rows[s:s + step]] - x[idx[s:s + step]]
    c0 += np.bincount(rows[s:s + step][(d * d).sum(axis=1) < 6.25], minlength=n)
def waste(v, label):
    m = (n // 64) * 64
    w = v[:m].reshape(-1, 64)
    print(f"{label}: mean {v.mean():.2f} sd {v.std():.2f}; per wavefront max/mean = {w.max(axis=1).sum() / w.mean(axis=1).sum():.4f}")
    b = v[: (n // 1024) * 1024].reshape(-1, 1024)
    s = -np.sort(-b, axis=1).reshape(-1, 16, 64)
    print(f"   sorted inside each 1024-bead workgroup: max/mean = {s.max(axis=2).sum() / s.mean(axis=2).sum():.4f}")
waste(nn.astype(np.int64), "row length")
waste(c0, "in-cutoff entries")
