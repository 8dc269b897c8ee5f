#!/usr/bin/env python3
"""How the border pass (ucg_border_count: sort of the owned beads + k_border_candidates) of ONE rank scales with the number of
ranks of the grid: 125 000 beads in the brick of rank 0 of a 1x1x1, 2x2x2 and 3x3x3 grid over a box of the matching size (what
each rank of an 8- / 27-GPU run does at every re-neighbouring).  usage: python tools/border_scaling.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def main():
    pkg = entry.load_package()
    beads = pkg.synth.make_beads(50, seed=3)  # 125 000 beads, box edge 53.86
    L = float(beads.boxhi[0] - beads.boxlo[0])
    for g in (1, 2, 3):
        ctx = pkg.capi.Context(0, dt=0.002)
        ctx.upload_beads(beads)
        ctx.domain_set(beads.boxlo, beads.boxlo + g * L, 2.5, 0.3, every=10, delay=0, check=1)
        ctx.decomp_set([g, g, g], 0)
        ctx.exchange_count()
        c = ctx.border_count()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            c = ctx.border_count()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"grid {g}x{g}x{g}: border_count {1e6 * dt:8.1f} us   ghosts sent {int(c.sum())} to {int((c > 0).sum())} ranks", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
