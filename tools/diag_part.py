import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import numpy as np
pkg = entry.load_package()
capi, synth = pkg.capi, pkg.synth
beads = synth.make_beads(50, seed=1)
deck = synth.make_deck(tempfile.mkdtemp(), "spline", 1024)
for slots in (1, 4, 8):
    ctx = capi.Context(0, dt=0.002)
    ctx.set_option("gather_slots", slots)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=10, delay=0, check=1)
    ctx.neigh_rebuild()
    pair = capi.Pair(ctx, "table_ucgld")
    pair.settings(deck.pair_style_args()); pair.coeff(deck.pair_coeff_args()); pair.init(2, 1.0)
    def tm(fn, n=50):
        fn(); ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        ctx.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    full = tm(lambda: pair.compute(0, 0))
    p1 = tm(lambda: pair.compute_part(1))
    p2 = tm(lambda: pair.compute_part(2))
    ctx.force_clear(); pair.compute_part(1); f = ctx.atoms_download()["f"]
    print(f"slots {slots}: full {full:.1f} us, part1 {p1:.1f} us, part2 {p2:.1f} us, interior beads {np.any(f != 0, axis=1).mean():.3f}")
    pair.close(); ctx.close()
