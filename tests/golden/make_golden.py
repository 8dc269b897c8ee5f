#!/usr/bin/env python3
"""Generate tests/golden/*.json.

The reference ships no fixtures and cannot be built here, so these are NOT reference
outputs: they are regression vectors of the CPU oracle in its canonical order (the order the GPU reproduces bit
for bit), frozen so that the oracle and the GPU path cannot drift together unnoticed.  The RANMAR entries are the
exception: the published check values of Marsaglia, Zaman & Tsang (1990) are external known answers.

    python tests/golden/make_golden.py        # rewrites the fixtures next to this file

Every float is stored as its IEEE-754 bit pattern (hex), inputs included, so the fixtures do not depend on any
random generator or on numpy versions."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402


def hexa(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return {"shape": list(a.shape), "f64hex": a.view(np.uint64).ravel().tolist()}


def inta(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return {"shape": list(a.shape), "i64": a.ravel().tolist()}


def beads_record(b):
    return dict(n=int(b.n), boxlo=hexa(b.boxlo), boxhi=hexa(b.boxhi), x=hexa(b.x), v=hexa(b.v), type=inta(b.type),
                tag=inta(b.tag), mask=inta(b.mask), ucgstate=inta(b.ucgstate), ucgl=hexa(b.ucgl), ucgvl=hexa(b.ucgvl),
                ucgml=hexa(b.ucgml), ucgp=hexa(b.ucgp), mass=hexa(b.mass), ntypes=int(b.ntypes))


def file_sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    pkg, orc = util.load_package(), util.load_oracle()
    out = {}
    # 1. RANMAR: published check values (seed ij = 1802, kl = 9373 -> values 20001..20006 x 2^24)
    out["ranmar_published"] = [6533892, 14220222, 7275067, 6172232, 8354498, 10633180]
    # 2. pair styles on 125 beads, canonical order
    beads = pkg.synth.make_beads(5, seed=2024)
    beads.ucgp = np.clip(np.random.default_rng(7).uniform(size=beads.n), 1e-6, 1 - 1e-6)
    cases = {}
    # sum_fixed: the summation mode of the library pair the vectors are for (include/ucg_hip.h, ucg_pair_sum_fixed): ordered
    # sums by default, fixed sums for the "*_vrow" cases (the same deck with option pair_vrow 1: the virtual-row kernels)
    for name, style, kw, extra, fixed in (("ucgld_spline1024", "table_ucgld", {}, (), False),
                                          ("ucgld_spline1024_vrow", "table_ucgld", {}, (), True),
                                          ("ucgld_linear2000", "table_ucgld", dict(tabstyle="linear", tablength=2000), (), False),
                                          ("bethe_pseudo_yes", "table_ucg_bethe", {}, ("pseudo", "yes"), False),
                                          ("bethe_pseudo_yes_vrow", "table_ucg_bethe", {}, ("pseudo", "yes"), True),
                                          ("bethe_mf", "table_ucg_bethe", {}, ("method", "mf"), False),
                                          ("density", "table_ucg_bethe_density", dict(density=(11.3, 1.5), extra11=0.05), (), False)):
        deck = util.make_deck(kw.get("tabstyle", "spline"), kw.get("tablength", 1024), extra_keywords=extra,
                              **{k: v for k, v in kw.items() if k in ("density", "extra11")})
        op = util.oracle_pair(style, deck)
        op.set_sum_fixed(fixed)
        sim = util.oracle_sim(beads, op, mode=1, dt=0.004, nve=True, every=1,
                              langevin=(1.0, 1.0, 1.0, 48279) if style == "table_ucgld" else None,
                              ucgstate="ld" if style == "table_ucgld" else ("mc", 4242, 0.3))
        assert sim.setup(10) == 0
        A0 = sim.arrays()
        e0 = sim.ev()["eng_vdwl"]
        assert sim.run(10, 0) == 0
        A1 = sim.arrays()
        cases[name] = dict(style=style, sum_fixed=fixed, pair_vrow=1 if name.endswith("_vrow") else 0, tabstyle=deck.tabstyle, tablength=deck.tablength, extra=list(extra),
                           deck_kw={k: v for k, v in kw.items() if k in ("density", "extra11")},
                           table_sha256=file_sha(deck.table_file),
                           setup=dict(tag=inta(A0["tag"]), f=hexa(A0["f"]), scores=hexa(A0["scores"]),
                                      ucgforce=hexa(A0["ucgforce"]), ucgp=hexa(A0["ucgp"]), eng_vdwl=hexa([e0])),
                           after10=dict(tag=inta(A1["tag"]), x=hexa(A1["x"]), v=hexa(A1["v"]), ucgl=hexa(A1["ucgl"]),
                                        ucgstate=inta(A1["ucgstate"])))
    out["beads"] = beads_record(beads)
    out["cases"] = cases
    with open(os.path.join(HERE, "ucg_golden.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote", os.path.join(HERE, "ucg_golden.json"), os.path.getsize(os.path.join(HERE, "ucg_golden.json")), "bytes")


if __name__ == "__main__":
    main()
