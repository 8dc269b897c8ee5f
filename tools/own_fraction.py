#!/usr/bin/env python3
"""own_fraction.py -- what fraction of a bead's neighbours (list radius / force cutoff) are beads of its own
workgroup block, for blocks of B consecutive beads of the Morton-sorted order?  (sizes the "own-block pairs once"
form of the gather kernel: DESIGN.md 4.1)   usage: python tools/own_fraction.py [ncell]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as e  # noqa: E402

pkg = e.load_package()
capi, synth = pkg.capi, pkg.synth
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 64
beads = synth.make_beads(ncell, seed=12345)
ctx = capi.Context(0, dt=0.002)
ctx.upload_beads(beads)
ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
ctx.neigh_rebuild()
il, nn, fi, ne = ctx.neigh_download()
A = ctx.atoms_download(with_ghosts=True)
x = A["x"]
n = len(nn)
row = np.repeat(np.arange(n), nn)
m = ne & 0x1FFFFFFF
d = x[row] - x[m]
rsq = (d * d).sum(axis=1)
incut = rsq < 6.25
print(f"{n} beads, {len(m)} full-list entries ({len(m) / n:.1f} per bead), in cutoff {incut.mean():.3f}")
for B in (256, 512, 768, 1024):
    own = (m < n) & (row // B == m // B)
    print(f"block of {B:5d} beads: own-block share of all entries {own.mean():.3f}, of in-cutoff entries {own[incut].mean():.3f}")
