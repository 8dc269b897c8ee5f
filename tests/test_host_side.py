"""CPU tier: the product's host side (input parsing, table construction, C-ABI surface)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.capi.lib()
    hdr = open(os.path.join(ROOT, "include", "ucg_hip.h")).read()
    declared = set(re.findall(r"\b(ucg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed from include/ucg_hip.h"
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libucg_hip.so does not export {sym}"
    assert declared == set(pkg.capi.SYMBOLS) | set(pkg.ucgio.SYMBOLS)
    assert lib.ucg_abi_version() == 1


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(pkg.capi.UcgError):
        pkg.capi.Context(-1)


@pytest.mark.parametrize("style", ["table_ucgld", "table_ucg_bethe"])
@pytest.mark.parametrize("tabstyle,tablength", [("spline", 1024), ("linear", 4096), ("lookup", 777), ("spline", 33),
                                                ("bitmap", 12), ("bitmap", 9)])
def test_host_tables_bit_identical_to_oracle(pkg, orc, style, tabstyle, tablength):
    deck = util.make_deck(tabstyle, tablength)
    o = util.oracle_pair(style, deck)
    p = pkg.capi.Pair(None, style)
    p.settings(deck.pair_style_args())
    p.coeff(deck.pair_coeff_args())
    p.init(2, 1.0)
    assert p.table_count() == 4
    for m in range(4):
        ti, tp = o.table_info(m), p.table_params(m)
        assert all(ti[k] == tp[k] for k in tp)
        for w in ("rsq", "e", "f", "de", "df", "e2", "f2", "drsq", "e2file", "f2file", "rfile", "efile", "ffile"):
            a, b = o.table_array(m, w), p.table_array(m, w)
            if a is None or len(a) == 0:
                assert b is None or len(b) == 0
            else:
                assert util.bits_equal(a, b), (m, w)
    assert np.array_equal(o.int_array("tabindex"), p.tabindex())
    assert p.init_one(1, 2) == 2.5 and p.cutforce == 2.5


def test_rsq_and_match_tables(pkg, orc):
    # RSQ file grid + linear N == file N + rhi == cut => "match": file values used verbatim
    deck = util.make_deck("linear", 500, n_file=500, rmode="RSQ")
    o = util.oracle_pair("table_ucgld", deck)
    p = pkg.capi.Pair(None, "table_ucgld")
    p.settings(deck.pair_style_args())
    p.coeff(deck.pair_coeff_args())
    p.init(2, 1.0)
    assert o.table_info(0)["match"] == 1
    for w in ("e", "f", "de", "df"):
        assert util.bits_equal(o.table_array(0, w), p.table_array(0, w))
    assert util.bits_equal(p.table_array(0, "e"), p.table_array(0, "efile"))


def test_bitmap_tables(pkg, orc):
    """tabstyle bitmap: 2^N bins addressed by the bits of (float) r^2 (UCG/pair_table_ucgld.cpp:1247-1340 with upstream
    Pair::init_bitmap restated).  Properties any correct restatement has: every r^2 in [inner^2, cut^2] lands in a bin
    whose lower edge is below it and whose fraction is in [0, 1); interpolation reproduces the tabulated function; a
    BITMAP section of the file (2^N entries in bit order) is used verbatim ("match")."""
    deck = util.make_deck("bitmap", 12)
    o = util.oracle_pair("table_ucgld", deck)
    p = pkg.capi.Pair(None, "table_ucgld")
    p.settings(deck.pair_style_args())
    p.coeff(deck.pair_coeff_args())
    p.init(2, 1.0)
    rsqt, drsq = p.table_array(0, "rsq"), p.table_array(0, "drsq")
    assert len(rsqt) == 4096 and p.table_params(0)["innersq"] == rsqt.min() and 0.36 <= rsqt.min() < 0.3605
    nmask, nshift = [int(v) for v in o.table_array(0, "bits")]
    rng = np.random.default_rng(5)
    r = rng.uniform(np.sqrt(rsqt.min()) + 1e-6, 2.5, 20000)
    r32 = (r * r).astype(np.float32)
    idx = (r32.view(np.int32) & nmask) >> nshift
    frac = (r32.astype(np.float64) - rsqt[idx]) * drsq[idx]
    assert frac.min() >= 0.0 and frac.max() < 1.0
    ref = pkg.synth.lj_energy_force(np.sqrt(r32.astype(np.float64)), 1.0, 1.0, 0.0)
    for k in range(0, 20000, 97):
        e, f = p.single(1, 1, float(r[k] * r[k]), 1.0)
        assert abs(e - ref[0][k]) <= 2e-4 * (abs(ref[0][k]) + 1.0)
    # below the smallest tabulated r^2: the reference's error
    with pytest.raises(pkg.capi.UcgError, match="inner"):
        p.single(1, 1, 0.3, 1.0)
    # a BITMAP section in the file
    deck2 = util.make_deck("bitmap", 10, n_file=1024, rmode="BITMAP")
    o2 = util.oracle_pair("table_ucgld", deck2)
    p2 = pkg.capi.Pair(None, "table_ucgld")
    p2.settings(deck2.pair_style_args())
    p2.coeff(deck2.pair_coeff_args())
    p2.init(2, 1.0)
    assert o2.table_info(0)["match"] == 1
    for w in ("rsq", "e", "f", "de", "df", "drsq", "rfile"):
        assert util.bits_equal(o2.table_array(0, w), p2.table_array(0, w)), w
    assert util.bits_equal(p2.table_array(0, "e"), p2.table_array(0, "efile"))
    # and the mismatch the reference rejects: bitmapped file, other table length
    deck3 = util.make_deck("bitmap", 11, n_file=1024, rmode="BITMAP")
    p3 = pkg.capi.Pair(None, "table_ucgld")
    p3.settings(deck3.pair_style_args())
    with pytest.raises(pkg.capi.UcgError, match="Bitmapped table in file does not match"):
        p3.coeff(deck3.pair_coeff_args())


def test_single_matches_table_eval(pkg, orc):
    deck = util.make_deck("spline", 1024)
    p = pkg.capi.Pair(None, "table_ucgld")
    p.settings(deck.pair_style_args())
    p.coeff(deck.pair_coeff_args())
    p.init(2, 1.0)
    e, f = p.single(1, 1, 1.0)  # r = sigma: u = 0, f/r = 24
    assert abs(e) < 1e-4 and abs(f - 24.0) < 1e-2
    e12, f12 = p.single(1, 2, 1.2599210498948732)  # r = 2^(1/6): minimum of eps_01 = 0.8
    assert abs(e12 + 0.8) < 1e-5 and abs(f12) < 1e-3
    with pytest.raises(pkg.capi.UcgError):
        p.single(1, 1, 0.1)
    with pytest.raises(pkg.capi.UcgError):
        p.single(1, 1, 6.3)


def test_input_errors_match_reference_messages(pkg):
    deck = util.make_deck("spline", 256)
    p = pkg.capi.Pair(None, "table_ucgld")
    with pytest.raises(pkg.capi.UcgError, match="Unknown table style"):
        p.settings(["cubic", "100", deck.conf_file])
    with pytest.raises(pkg.capi.UcgError, match="Illegal number of pair table entries"):
        p.settings(["spline", "1", deck.conf_file])
    with pytest.raises(pkg.capi.UcgError, match="Cannot open file"):
        p.settings(["spline", "100", "/no/such/file"])
    p.settings(deck.pair_style_args())
    with pytest.raises(pkg.capi.UcgError, match="Incorrect number of arguments"):
        p.coeff(deck.pair_coeff_args()[:-3])
    bad = deck.pair_coeff_args()
    bad[6] = "3.5"
    with pytest.raises(pkg.capi.UcgError, match="cutoff outside of table"):
        p.coeff(bad)
    bad = deck.pair_coeff_args()
    bad[5] = "NOPE"
    with pytest.raises(pkg.capi.UcgError, match="Did not find keyword"):
        p.coeff(bad)
    p.settings(deck.pair_style_args())
    p.coeff(deck.pair_coeff_args())
    with pytest.raises(pkg.capi.UcgError, match="t_target"):
        p.init(2, 0.0)
    with pytest.raises(pkg.capi.UcgError, match="Unknown pair_style table keyword"):
        p.settings(deck.pair_style_args() + ["bogus"])


def test_bethe_keywords(pkg, orc):
    deck = util.make_deck("linear", 300, extra_keywords=("method", "mf", "pseudo", "no", "prior", "chemical_potential"))
    o = util.oracle_pair("table_ucg_bethe", deck, T=0.7)
    p = pkg.capi.Pair(None, "table_ucg_bethe")
    p.settings(deck.pair_style_args())
    p.coeff(deck.pair_coeff_args())
    p.init(2, 0.7)
    pri = o.dbl_array("prior_prob_from_type").reshape(2, 2)[1]
    assert abs(pri.sum() - 1.0) < 1e-15 and pri[0] > pri[1]
    with pytest.raises(pkg.capi.UcgError, match="please write mf or bethe"):
        p.settings(deck.pair_style_args()[:3] + ["method", "xyz"])


def test_synthetic_inputs_follow_data_atom_post(pkg):
    b = pkg.synth.make_beads(4)
    assert b.n == 64 and np.all((b.ucgl >= 0) & (b.ucgl <= 1)) and set(np.unique(b.ucgstate)) <= {0, 1}
    assert np.all(b.ucgp == -1.0) and np.all((b.x >= 0) & (b.x < b.boxhi))
    assert abs(b.n / np.prod(b.boxhi) - 0.8) < 1e-12
    f = pkg.synth.make_beads(3, lattice="fcc")
    assert f.n == 108
