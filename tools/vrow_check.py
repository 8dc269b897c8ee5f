#!/usr/bin/env python3
"""development aid: the virtual-row kernel (option pair_vrow) against the plain gather kernel on the same beads"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
synth, capi = pkg.synth, pkg.capi


def run(style, extra, ncell, vrow, steps=0):
    deck = synth.make_deck(tempfile.mkdtemp(prefix="vrow_"), "spline", 1024, extra_keywords=extra)
    beads = synth.make_beads(ncell, seed=7)
    rng = np.random.default_rng(3)
    beads.ucgp = np.clip(rng.uniform(size=beads.n), 1e-6, 1 - 1e-6)
    ctx = capi.Context(-1, dt=0.002)
    ctx.set_option("pair_vrow", vrow)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=2, delay=0, check=1)
    ctx.neigh_rebuild()
    gp = capi.Pair(ctx, style)
    gp.settings(deck.pair_style_args())
    gp.coeff(deck.pair_coeff_args())
    gp.init(2, 1.0)
    e, v = gp.compute(1, 1)
    gp.check_errors()
    A = ctx.atoms_download()
    gp.compute(0, 0)
    B = ctx.atoms_download()
    out = dict(e=e, v=v, f=A["f"], uf=A["ucgforce"], s=A["scores"], f2=B["f"], tag=A["tag"])
    if steps:
        ctx.fix_ucgstate("ld" if style == "table_ucgld" else None)
        if style == "table_ucgld":
            ctx.fix_ucgld_langevin(1.0, 1.0, 1.0, 48279)
        ctx.md_attach(gp, nve=True, langevin=style == "table_ucgld", ucgstate=True)
        ctx.md_setup(steps)
        ctx.md_run(steps, 0)
        gp.check_errors()
        C = ctx.atoms_download()
        out.update(x=C["x"], l=C["ucgl"], tag2=C["tag"], st=C["ucgstate"], nrebuild=ctx.md_info()["nrebuild"])
    gp.close()
    ctx.close()
    return out


for style, extra in (("table_ucgld", ()), ("table_ucg_bethe", ()), ("table_ucg_bethe", ("pseudo", "no"))):
    for ncell in (9, 20):
        a, b = run(style, extra, ncell, 0, 30), run(style, extra, ncell, 1, 30)
        assert np.array_equal(a["tag"], b["tag"])
        rel = lambda x, y: float(np.abs(x - y).max() / np.abs(x).max())
        print(style, extra, ncell, "f", rel(a["f"], b["f"]), "uf", float(np.abs(a["uf"] - b["uf"]).max()), "scores", rel(a["s"], b["s"]),
              "E", abs(a["e"] - b["e"]) / abs(a["e"]), "vir", rel(a["v"], b["v"]), "noEV==EV", bool(np.array_equal(b["f"], b["f2"])),
              "traj dx", float(np.abs(a["x"] - b["x"]).max()), "states differ", int((a["st"] != b["st"]).sum()), "rebuilds", a["nrebuild"], b["nrebuild"])
