"""Committed regression vectors (tests/golden/ucg_golden.json, made by tests/golden/make_golden.py).

They are oracle outputs in the canonical order, not reference outputs (the reference has no fixtures and cannot
be built here); the RANMAR check values are the published ones.  CPU tier: the oracle still produces them bit
for bit.  GPU tier: so does the HIP path."""
import hashlib
import json
import os

import numpy as np
import pytest

import util

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    with open(os.path.join(HERE, "golden", "ucg_golden.json")) as fh:
        return json.load(fh)


def _f(rec):
    return np.array(rec["f64hex"], dtype=np.uint64).view(np.float64).reshape(rec["shape"])


def _i(rec):
    return np.array(rec["i64"], dtype=np.int64).reshape(rec["shape"])


def _beads(pkg, g):
    b = g["beads"]
    return pkg.synth.Beads(n=b["n"], boxlo=_f(b["boxlo"]), boxhi=_f(b["boxhi"]), x=_f(b["x"]), v=_f(b["v"]),
                           type=_i(b["type"]).astype(np.int32), tag=_i(b["tag"]).astype(np.int32),
                           mask=_i(b["mask"]).astype(np.int32), ucgstate=_i(b["ucgstate"]).astype(np.int32),
                           ucgl=_f(b["ucgl"]), ucgvl=_f(b["ucgvl"]), ucgml=_f(b["ucgml"]), ucgp=_f(b["ucgp"]),
                           mass=_f(b["mass"]), ntypes=b["ntypes"])


def _deck(case):
    kw = dict(case["deck_kw"])
    if "density" in kw:
        kw["density"] = tuple(kw["density"])
    return util.make_deck(case["tabstyle"], case["tablength"], extra_keywords=tuple(case["extra"]), **kw)


def _fixes(case):
    ucgld = case["style"] == "table_ucgld"
    return ((1.0, 1.0, 1.0, 48279) if ucgld else None), ("ld" if ucgld else ("mc", 4242, 0.3))


CASES = ["ucgld_spline1024", "ucgld_spline1024_vrow", "ucgld_linear2000", "bethe_pseudo_yes", "bethe_pseudo_yes_vrow",
         "bethe_mf", "density"]


def test_ranmar_known_answers_from_the_fixture(orc):
    # the published check values (external known answers), read from the fixture file
    g = _load()
    L = orc.lib()
    r = orc.RanMars()
    L.orc_ranmars_init(r, 1802 * 30082 + 9373 + 1)
    for _ in range(19999):
        L.orc_ranmars_uniform(r)
    assert [int(L.orc_ranmars_uniform(r) * 4096.0 * 4096.0) for _ in range(6)] == g["ranmar_published"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(pkg, orc, name):
    g = _load()
    case = g["cases"][name]
    beads = _beads(pkg, g)
    deck = _deck(case)
    assert hashlib.sha256(open(deck.table_file, "rb").read()).hexdigest() == case["table_sha256"]
    lang, ucgst = _fixes(case)
    op = util.oracle_pair(case["style"], deck)
    op.set_sum_fixed(case["sum_fixed"])
    sim = util.oracle_sim(beads, op, mode=1, dt=0.004, nve=True, every=1, langevin=lang, ucgstate=ucgst)
    assert sim.setup(10) == 0
    A = sim.arrays()
    assert np.array_equal(A["tag"], _i(case["setup"]["tag"]))
    for k in ("f", "scores", "ucgforce", "ucgp"):
        assert util.bits_equal(A[k], _f(case["setup"][k])), k
    assert util.bits_equal([sim.ev()["eng_vdwl"]], _f(case["setup"]["eng_vdwl"]))
    assert sim.run(10, 0) == 0
    A = sim.arrays()
    assert np.array_equal(A["tag"], _i(case["after10"]["tag"]))
    for k in ("x", "v", "ucgl"):
        assert util.bits_equal(A[k], _f(case["after10"][k])), k
    assert np.array_equal(A["ucgstate"], _i(case["after10"]["ucgstate"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_golden(fresh_ctx, pkg, name):
    g = _load()
    case = g["cases"][name]
    beads = _beads(pkg, g)
    deck = _deck(case)
    lang, ucgst = _fixes(case)
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.004)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    ctx.set_option("pair_vrow", case["pair_vrow"])
    gp = util.gpu_pair(ctx, case["style"], deck)
    assert gp.sum_fixed == case["sum_fixed"]  # the summation mode the vectors were made in
    if lang:
        ctx.fix_ucgld_langevin(*lang)
    if ucgst == "ld":
        ctx.fix_ucgstate("ld")
    else:
        ctx.fix_ucgstate("mc", ucgst[1], ucgst[2])
    ctx.md_attach(gp, nve=True, langevin=lang is not None, ucgstate=True)
    ctx.md_setup(10)
    A = ctx.atoms_download()
    assert np.array_equal(A["tag"], _i(case["setup"]["tag"]))
    for k in ("f", "scores", "ucgforce", "ucgp"):
        assert util.bits_equal(A[k], _f(case["setup"][k])), k
    e = ctx.md_thermo()["eng_vdwl"]
    e0 = float(_f(case["setup"]["eng_vdwl"])[0])
    assert abs(e - e0) <= 1e-12 * abs(e0)  # block-wise energy reduction: tolerance, not bits
    ctx.md_run(10, 0)
    gp.check_errors()
    A = ctx.atoms_download()
    assert np.array_equal(A["tag"], _i(case["after10"]["tag"]))
    for k in ("x", "v", "ucgl"):
        assert util.bits_equal(A[k], _f(case["after10"][k])), k
    assert np.array_equal(A["ucgstate"], _i(case["after10"]["ucgstate"]))
