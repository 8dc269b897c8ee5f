#!/usr/bin/env python3
"""How long does the bench workload run before lambda (never clamped by fix nve/ucgld) drives a pair inside the
table's inner cutoff?  Prints lambda statistics every 250 steps.  usage: stability.py [nve|wall] [nsteps]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
capi, synth = pkg.capi, pkg.synth
kind = sys.argv[1] if len(sys.argv) > 1 else "nve"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
beads = synth.make_beads(40, seed=12345)
deck = synth.make_deck(tempfile.mkdtemp(), "spline", 1024)
ctx = capi.Context(0, dt=0.002)
ctx.upload_beads(beads)
ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
pair = capi.Pair(ctx, "table_ucgld")
pair.settings(deck.pair_style_args()); pair.coeff(deck.pair_coeff_args()); pair.init(2, 1.0)
ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279); ctx.fix_ucgstate("ld")
if kind == "wall":
    ctx.fix_nve_ucgld_wall_hard(False, 0.1)
ctx.md_attach(pair, nve="wall" if kind == "wall" else True, langevin=True, ucgstate=True)
ctx.md_setup(nsteps)
done = 0
while done < nsteps:
    ctx.md_run(250, 0)
    done += 250
    try:
        pair.check_errors()
    except Exception as e:  # noqa: BLE001
        print(f"step {done}: {e}")
        break
    A = ctx.atoms_download()
    l = A["ucgl"]
    print(f"step {done}: lambda mean {l.mean():+.3f} min {l.min():+.3f} max {l.max():+.3f}  |v| max {np.abs(A['v']).max():.2f}  ucgp mean {A['ucgp'].mean():.3f}")
