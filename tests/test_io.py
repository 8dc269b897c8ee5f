"""On-disk formats of atom style ucg: csrc/ucg_io.cpp against the Python restatement
oracle/orc_io.py and a hand-checked golden dump (tests/golden/dump_ucg_golden.txt).  Host code: CPU tier, plus one
GPU test that dumps from the resident loop and reads the snapshot back."""
import os
import sys

import numpy as np
import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc_io  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "dump_ucg_golden.txt")


def small_atoms(n=57, seed=3):
    rng = np.random.default_rng(seed)
    lo, hi = np.array([-1.5, 0.0, 2.0]), np.array([6.5, 9.0, 12.25])
    ids = rng.permutation(n).astype(np.int32) + 1
    a = dict(id=ids, type=rng.integers(1, 3, n).astype(np.int32), molecule=((ids - 1) // 4 + 1).astype(np.int32),
             x=lo + rng.random((n, 3)) * (hi - lo), v=rng.normal(size=(n, 3)), f=rng.normal(size=(n, 3)) * 1e3,
             q=rng.normal(size=n) * 0.1, image=rng.integers(-2, 3, (n, 3)).astype(np.int32),
             ucgstate=rng.integers(0, 2, n).astype(np.int32), ucgl=rng.random(n), ucgvl=rng.normal(size=n) * 1e-3,
             ucgml=np.full(n, 10.0), ucgp=rng.random(n), ucgforce=rng.normal(size=n), mass=np.array([0.0, 1.0, 2.5]),
             boxlo=lo, boxhi=hi, ntypes=2)
    a["ucgl"][:3] = [0.0, 1.0, 1e-300]
    a["ucgp"][:3] = [1e-6, 1 - 1e-6, 0.5]
    return a


def test_io_symbols_are_declared_and_exported(pkg):
    lib = pkg.capi.lib()
    hdr = open(os.path.join(ROOT, "include", "ucg_hip.h")).read()
    for s in pkg.ucgio.SYMBOLS:
        assert hasattr(lib, s) and (s + "(") in hdr


@pytest.mark.parametrize("columns,kw", [
    ("id type x y z ucgstate ucgl ucgp", {}),
    ("id mol type mass xs ys zs xu yu zu ix iy iz vx vy vz fx fy fz q ucgstate ucgl ucgp ucgvl ucgml ucgforce", {}),
    ("id ucgl ucgp", dict(thresh=[("ucgl", ">", "0.5")])),
    ("id ucgstate ucgl", dict(thresh=[("ucgstate", "==", "1"), ("ucgp", "<=", "0.75")], sort_id=True)),
    ("id type x ucgl", dict(sort_id=True, fmt_float="%20.15g", fmt_int="%6d")),
])
def test_dump_text_matches_oracle_bytes(pkg, tmp_path, columns, kw):
    a = small_atoms()
    want = orc_io.dump_text(a, columns, timestep=4200, **kw)
    modify = ["thresh %s %s %s" % t for t in kw.get("thresh", ())]
    if kw.get("sort_id"):
        modify.append("sort id")
    if kw.get("fmt_float"):
        modify += ["format float " + kw["fmt_float"], "format int " + kw["fmt_int"]]
    p = tmp_path / "d.dump"
    nw = pkg.ucgio.write_dump(p, a, columns, timestep=4200, modify=modify)
    got = open(p).read()
    assert got == want
    assert nw == int(want.split("\n")[3])


def test_dump_modify_format_line_column_and_boundary(pkg, tmp_path):
    """dump_modify format line "..." / format M / boundary: the per-column formats are the words of the line format,
    overridden column by column (DumpCustom::init_style, dump_custom.cpp:262-290)"""
    a = small_atoms(n=5)
    p = tmp_path / "f.dump"
    pkg.ucgio.write_dump(p, a, "id type ucgl ucgp", timestep=3,
                         modify=['format line "%d %3d %10.4f %g"', "format 4 %.3e", "boundary pp pp ff", "sort id"])
    lines = open(p).read().split("\n")
    assert lines[4] == "ITEM: BOX BOUNDS pp pp ff"
    order = np.argsort(a["id"])
    for ln, i in zip(lines[9:14], order):
        assert ln == "%d %3d %10.4f %.3e" % (a["id"][i], a["type"][i], a["ucgl"][i], a["ucgp"][i])
    with pytest.raises(pkg.capi.UcgError, match="format line is too short"):
        pkg.ucgio.write_dump(p, a, "id type ucgl ucgp", modify=['format line "%d %d"'])
    with pytest.raises(pkg.capi.UcgError, match="Invalid dump_modify thresh operator"):
        pkg.ucgio.write_dump(p, a, "id", modify=["thresh ucgl => 0.5"])
    with pytest.raises(pkg.capi.UcgError, match="not supported"):
        pkg.ucgio.write_dump(p, a, "id", modify=["pbc yes"])


def test_dump_golden_file(pkg, tmp_path):
    """six atoms written by hand in the native format (header widths, "%d" / "%g" columns, one blank between columns,
    none at the end of a line); both the oracle and the library must reproduce the file byte for byte"""
    a = dict(id=np.array([3, 1, 2, 6, 5, 4], np.int32), type=np.array([1, 2, 1, 2, 1, 1], np.int32),
             x=np.array([[0.5, 1.0, 1.5], [2.25, 0.125, 3.0], [1e-5, 9.999999, 4.0], [7.0, 8.0, 9.0],
                         [1234567.0, 0.1, 0.2], [3.0, 3.0, 3.0]]),
             ucgstate=np.array([0, 1, 1, 0, 1, 0], np.int32),
             ucgl=np.array([0.0, 1.0, 0.5, 0.123456789, 1e-7, 0.75]),
             ucgp=np.array([1e-6, 0.999999, 0.5, -1.0, 0.25, 0.3333333333]),
             boxlo=np.zeros(3), boxhi=np.array([10.0, 10.0, 10.0]), ntypes=2)
    want = open(GOLDEN).read()
    assert orc_io.dump_text(a, "id type x y z ucgstate ucgl ucgp", timestep=100) == want
    p = tmp_path / "g.dump"
    pkg.ucgio.write_dump(p, a, "id type x y z ucgstate ucgl ucgp", timestep=100)
    assert open(p).read() == want
    # and the reader understands it
    snap = pkg.ucgio.load_dump(GOLDEN, 100)
    assert snap["columns"] == "id type x y z ucgstate ucgl ucgp".split() and snap["natoms"] == 6
    assert np.array_equal(snap["values"][:, 0], a["id"]) and np.array_equal(snap["values"][:, 5], a["ucgstate"])


def test_dump_append_scan_and_load(pkg, tmp_path):
    a = small_atoms()
    p = tmp_path / "traj.dump"
    for k, ts in enumerate((0, 50, 100)):
        a["ucgl"] = (a["ucgl"] + 0.25) % 1.0
        pkg.ucgio.write_dump(p, a, "id ucgstate ucgl ucgp x y z", timestep=ts, append=k > 0,
                             modify=["format float %.17g"] + (["thresh ucgl < 0.5"] if ts == 50 else []))
    snaps = pkg.ucgio.dump_snapshots(p)
    assert [s[0] for s in snaps] == [0, 50, 100] and snaps[0][1] == snaps[2][1] == 57 and snaps[1][1] < 57
    ref = orc_io.parse_dump(open(p).read())
    for ts, r in zip((0, 50, 100), ref):
        s = pkg.ucgio.load_dump(p, ts)
        assert s["timestep"] == ts and s["columns"] == r["columns"]
        assert util.bits_equal(s["values"], r["values"]) and util.bits_equal(s["boxlo"], r["boxlo"])
    last = pkg.ucgio.load_dump(p, 100)
    order = np.argsort(last["values"][:, 0])
    assert util.bits_equal(last["values"][order, 2], a["ucgl"][np.argsort(a["id"])])  # %.17g round-trips every bit
    with pytest.raises(pkg.capi.UcgError, match="does not contain requested snapshot"):
        pkg.ucgio.load_dump(p, 75)


@pytest.mark.parametrize("options,kw", [("", {}), ("box no", dict(box=False)), ("trim yes", dict(trim=True)),
                                        ("wrapped no", dict(wrapped=False)), ("replace no trim yes", dict(replace=False, trim=True))])
def test_read_dump_matches_oracle(pkg, tmp_path, options, kw):
    a = small_atoms()
    b = small_atoms(seed=11)  # other values, other ID order
    keep = np.sort(np.random.default_rng(5).permutation(57)[:40])
    sub = {k: (v[keep] if isinstance(v, np.ndarray) and k not in ("boxlo", "boxhi", "mass") else v) for k, v in b.items()}
    sub["boxhi"] = b["boxhi"] * 1.25
    p = tmp_path / "s.dump"
    pkg.ucgio.write_dump(p, sub, "id type xs ys zs ix iy iz vx vy vz ucgstate ucgl ucgp", timestep=7, modify=["format float %.17g"])
    fields = "x y z ix iy iz vx vy vz ucgstate ucgl ucgp"
    snap = orc_io.parse_dump(open(p).read())[0]
    want, wstat = orc_io.read_dump(snap, fields, a, **kw)
    got, gstat = pkg.ucgio.read_dump(p, 7, fields, a, options)
    assert gstat == wstat and gstat["snapshot"] == 40
    assert gstat["natoms"] == (40 if "trim yes" in options else 57)
    for k in ("id", "type", "molecule", "ucgstate", "image"):
        assert np.array_equal(got[k], want[k]), k
    for k in ("x", "v", "f", "q", "ucgl", "ucgvl", "ucgml", "ucgp", "boxlo", "boxhi"):
        assert util.bits_equal(got[k], want[k]), k
    assert np.array_equal(a["ucgl"], small_atoms()["ucgl"])  # the caller's arrays are not written


def test_read_dump_coordinate_representations(pkg, tmp_path):
    """x may come as x / xs / xu / xsu in the file (reader_native.cpp:329-398); `scaled` / `wrapped` say which one is
    preferred when several are present, and unwrapped input resets the image flags (read_dump.cpp:917)"""
    a = small_atoms(n=9)
    prd = a["boxhi"] - a["boxlo"]
    p = tmp_path / "c.dump"
    pkg.ucgio.write_dump(p, a, "id x y z xs ys zs xu yu zu xsu ysu zsu ix iy iz", modify=["format float %.17g"])
    base = dict(a, x=np.zeros_like(a["x"]))
    got, _ = pkg.ucgio.read_dump(p, 0, "x y z", base)                      # wrapped, unscaled: the x columns
    assert util.bits_equal(got["x"], a["x"]) and np.array_equal(got["image"], a["image"])
    got, _ = pkg.ucgio.read_dump(p, 0, "x y z", base, "scaled yes")        # xs * prd + lo
    xs = (a["x"] - a["boxlo"]) * (1.0 / prd)
    assert util.bits_equal(got["x"], xs * prd + a["boxlo"])
    got, _ = pkg.ucgio.read_dump(p, 0, "x y z", base, "wrapped no")        # xu as is, images zeroed
    assert util.bits_equal(got["x"], a["x"] + a["image"] * prd) and np.all(got["image"] == 0)
    got, _ = pkg.ucgio.read_dump(p, 0, "x y z ix iy iz", base, "wrapped no")
    assert np.all(got["image"] == 0)
    # only scaled columns in the file: they are taken whatever the preference
    q = tmp_path / "s.dump"
    pkg.ucgio.write_dump(q, a, "id xs ys zs", modify=["format float %.17g"])
    got, _ = pkg.ucgio.read_dump(q, 0, "x y z", base)
    assert util.bits_equal(got["x"], xs * prd + a["boxlo"])


def test_read_dump_errors(pkg, tmp_path):
    a = small_atoms()
    p = tmp_path / "e.dump"
    pkg.ucgio.write_dump(p, a, "id x y z ucgl")
    with pytest.raises(pkg.capi.UcgError, match="field not found in dump file: ucgp"):
        pkg.ucgio.read_dump(p, 0, "x ucgp", a)
    with pytest.raises(pkg.capi.UcgError, match="not supported for atom style ucg"):
        pkg.ucgio.read_dump(p, 0, "x", a, "purge yes add yes")
    noucg = {k: v for k, v in a.items() if k != "ucgl"}
    with pytest.raises(pkg.capi.UcgError, match="UCG L property that isn't supported by atom style"):
        pkg.ucgio.read_dump(p, 0, "ucgl", noucg)
    with pytest.raises(pkg.capi.UcgError, match="Invalid attribute"):
        pkg.ucgio.write_dump(p, a, "id msucgl")
    with pytest.raises(pkg.capi.UcgError, match="Cannot open"):
        pkg.ucgio.load_dump(tmp_path / "missing.dump")


def test_data_file_round_trip_and_data_atom_post(pkg, tmp_path):
    a = small_atoms()
    a["ucgl"][5:8] = [-0.25, 1.75, 0.5]     # outside [0, 1]: clamped on read (data_atom_post)
    a["ucgstate"][5:8] = [-3, 7, 1]
    p = tmp_path / "sys.data"
    pkg.ucgio.write_data(p, a, timestep=12)
    b = pkg.ucgio.read_data(p)
    st, lam, post = orc_io.data_post(a["ucgstate"], a["ucgl"])
    assert b["n"] == 57 and b["ntypes"] == 2
    for k in ("id", "type", "molecule", "image"):
        assert np.array_equal(b[k], a[k]), k
    for k in ("x", "v", "q", "ucgvl", "ucgml", "mass", "boxlo", "boxhi"):
        assert util.bits_equal(b[k], a[k]), k
    assert np.array_equal(b["ucgstate"], st) and util.bits_equal(b["ucgl"], lam) and util.bits_equal(b["ucgp"], post)
    # a hand-written file in the documented column order, with comments and without image flags / velocities
    q = tmp_path / "hand.data"
    q.write_text("""LAMMPS data file, atom_style ucg

3 atoms
2 atom types
0 bonds

0.0 4.0 xlo xhi
0.0 4.0 ylo yhi   # comment
-1.0 3.0 zlo zhi

Masses

1 1.0
2 3.5  # heavy

Atoms # ucg

2 1 2 0.0 1.0 2.0 0.5 1 0.75 10.0
1 1 1 -0.5 0.25 0.5 0.75 0 1.5 12.0
3 2 1 0.0 3.0 3.0 -0.5 5 -0.1 10.0
""")
    h = pkg.ucgio.read_data(q)
    assert h["id"].tolist() == [2, 1, 3] and h["type"].tolist() == [2, 1, 1] and h["molecule"].tolist() == [1, 1, 2]
    assert h["ucgstate"].tolist() == [1, 0, 1] and h["ucgl"].tolist() == [0.75, 1.0, 0.0] and h["ucgp"].tolist() == [-1.0] * 3
    assert h["ucgml"].tolist() == [10.0, 12.0, 10.0] and h["mass"].tolist() == [0.0, 1.0, 3.5] and h["q"][1] == -0.5
    assert h["boxlo"].tolist() == [0.0, 0.0, -1.0] and np.all(h["v"] == 0.0) and np.all(h["image"] == 0)
    bad = tmp_path / "bad.data"
    bad.write_text(q.read_text().replace("2 1 2 0.0 1.0 2.0 0.5 1 0.75 10.0", "2 1 2 0.0 1.0 2.0 0.5 1 0.75"))
    with pytest.raises(pkg.capi.UcgError, match="Incorrect atom format"):
        pkg.ucgio.read_data(bad)


def test_restart_round_trip(pkg, tmp_path):
    a = small_atoms()
    p = tmp_path / "r.ucgrst"
    pkg.ucgio.write_restart(p, a, timestep=31337)
    b = pkg.ucgio.read_restart(p)
    assert b["timestep"] == 31337 and b["n"] == 57
    for k in ("id", "type", "molecule", "image", "ucgstate"):
        assert np.array_equal(b[k], a[k]), k
    for k in ("x", "v", "q", "ucgl", "ucgml", "ucgvl", "ucgp", "mass", "boxlo", "boxhi"):
        assert util.bits_equal(b[k], a[k]), k
    (tmp_path / "junk").write_bytes(b"not a restart file at all, but long enough to hold a header " * 4)
    with pytest.raises(pkg.capi.UcgError, match="UCGRST01"):
        pkg.ucgio.read_restart(tmp_path / "junk")


@pytest.mark.gpu
def test_resident_loop_dump_restart_and_resume(fresh_ctx, pkg, tmp_path):
    """run 10 steps, dump and write a restart; the dump read back through read_dump equals the device state bit for
    bit, and a fresh system started from the restart file has the same forces at that configuration (to rounding: its
    neighbour rows are ordered by the new positions)"""
    beads = pkg.synth.make_beads(12, seed=77)
    deck = util.make_deck("spline", 1024)
    ctx = fresh_ctx
    ctx.set_units(1.0, 1.0, 1.0, 0.002)
    ctx.upload_beads(beads)
    ctx.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
    gp = util.gpu_pair(ctx, "table_ucgld", deck)
    ctx.fix_ucgstate("ld")
    ctx.md_attach(gp, nve=True, langevin=False, ucgstate=True)  # plain nve/ucgld: final_integrate leaves lambda alone
    ctx.md_setup(10)
    ctx.md_run(10, 0)
    gp.check_errors()
    at = pkg.ucgio.atoms_of(ctx, beads.boxlo, beads.boxhi, ntypes=beads.ntypes, mass=beads.mass)
    cols = "id type x y z vx vy vz fx fy fz ucgstate ucgl ucgp"
    pkg.ucgio.write_dump(tmp_path / "t.dump", at, cols, timestep=10, modify=["sort id", "format float %.17g"])
    pkg.ucgio.write_restart(tmp_path / "t.rst", at, timestep=10)
    # read_dump into the initial configuration: every dumped field arrives bit for bit
    init = dict(id=beads.tag, type=beads.type, x=beads.x.copy(), v=beads.v.copy(), f=np.zeros_like(beads.x),
                ucgstate=beads.ucgstate, ucgl=beads.ucgl, ucgp=beads.ucgp, boxlo=beads.boxlo, boxhi=beads.boxhi, ntypes=beads.ntypes)
    got, stat = pkg.ucgio.read_dump(tmp_path / "t.dump", 10, "x y z vx vy vz fx fy fz ucgstate ucgl ucgp", init)
    assert stat["replaced"] == beads.n
    by_id = np.argsort(at["id"])
    assert np.array_equal(got["id"], beads.tag)
    back = np.argsort(np.argsort(beads.tag))
    for k in ("x", "v", "f", "ucgl", "ucgp"):
        assert util.bits_equal(got[k], at[k][by_id][back]), k
    assert np.array_equal(got["ucgstate"], at["ucgstate"][by_id][back])
    # restart: a fresh system from the container gives the same forces at that configuration
    r = pkg.ucgio.read_restart(tmp_path / "t.rst")
    import copy
    b2 = copy.copy(beads)
    b2.x, b2.v, b2.type, b2.tag = r["x"], r["v"], r["type"], r["id"]
    b2.ucgstate, b2.ucgl, b2.ucgvl, b2.ucgml, b2.ucgp = r["ucgstate"], r["ucgl"], r["ucgvl"], r["ucgml"], r["ucgp"]
    b2.mask = np.ones(beads.n, np.int32)
    ctx2 = pkg.capi.Context(-1)
    try:
        ctx2.set_units(1.0, 1.0, 1.0, 0.002)
        ctx2.upload_beads(b2)
        ctx2.domain_set(beads.boxlo, beads.boxhi, 2.5, 0.3, every=1, delay=0, check=1)
        gp2 = util.gpu_pair(ctx2, "table_ucgld", deck)
        ctx2.md_attach(gp2, nve=False, langevin=False, ucgstate=False)
        ctx2.md_setup(0)
        gp2.check_errors()
        a2 = ctx2.atoms_download()
        o2, o1 = np.argsort(a2["tag"][:beads.n]), by_id
        # same configuration, but this list is built AT the dumped positions while the first run's list dates from its
        # last re-neighbouring: the rows hold the same pairs in another order, so the sums agree to rounding only
        assert np.abs(a2["f"][o2] - at["f"][o1]).max() <= 1e-11 * np.abs(at["f"]).max()
        assert np.abs(a2["ucgforce"][o2] - at["ucgforce"][o1]).max() <= 1e-11 * np.abs(at["ucgforce"]).max()
    finally:
        ctx2.close()
