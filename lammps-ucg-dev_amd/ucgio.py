"""ctypes binding of the ucg_io_* entry points (include/ucg_hip.h, csrc/ucg_io.cpp): the on-disk formats of atom
style ucg -- native text dumps with the ucgstate / ucgl / ucgp keywords, read_dump, data files, restart container
(SURVEY.md section 8 row f4).  Host code: no GPU, no context.  The functions are named after the LAMMPS commands
they stand for and take their arguments in the same words:

    write_dump(path, atoms, "id type x y z ucgstate ucgl ucgp", modify=["thresh ucgl > 0.5", "sort id"])
    read_dump(path, timestep, "x y z ucgstate ucgl ucgp", atoms, "box yes trim no")
    write_data / read_data, write_restart / read_restart

`atoms` is a dict of numpy arrays (the layout of Context.atoms_download(): tag/id, type, x, v, f, ucgstate, ucgl,
ucgvl, ucgml, ucgp, ucgforce; optional molecule, q, image, mass) plus boxlo / boxhi / ntypes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

SYMBOLS = [
    "ucg_io_dump_write", "ucg_io_dump_scan", "ucg_io_dump_header", "ucg_io_dump_load", "ucg_io_read_dump",
    "ucg_io_write_data", "ucg_io_data_header", "ucg_io_read_data", "ucg_io_write_restart", "ucg_io_restart_header",
    "ucg_io_read_restart",
]

_ip, _dp = C.POINTER(C.c_int), C.POINTER(C.c_double)


class IoAtoms(C.Structure):
    _fields_ = [("n", C.c_longlong), ("ntypes", C.c_int), ("boxlo", C.c_double * 3), ("boxhi", C.c_double * 3),
                ("id", _ip), ("type", _ip), ("molecule", _ip), ("ucgstate", _ip), ("image", _ip),
                ("x", _dp), ("v", _dp), ("f", _dp), ("q", _dp), ("ucgl", _dp), ("ucgvl", _dp), ("ucgml", _dp),
                ("ucgp", _dp), ("ucgforce", _dp), ("mass", _dp)]


INT_FIELDS = ("id", "type", "molecule", "ucgstate", "image")
DBL_FIELDS = ("x", "v", "f", "q", "ucgl", "ucgvl", "ucgml", "ucgp", "ucgforce", "mass")
_WIDTH = {"x": 3, "v": 3, "f": 3, "image": 3}
_SETUP = False


def _lib():
    global _SETUP
    L = capi.lib()
    if not _SETUP:
        cp, ll, llp = C.c_char_p, C.c_longlong, C.POINTER(C.c_longlong)
        ap = C.POINTER(IoAtoms)
        L.ucg_io_dump_write.argtypes = [cp, C.c_int, ll, ap, cp, cp, llp, cp, C.c_int]
        L.ucg_io_dump_scan.argtypes = [cp, llp, llp, C.c_int, _ip, cp, C.c_int]
        L.ucg_io_dump_header.argtypes = [cp, ll, llp, llp, _dp, _dp, cp, C.c_int, cp, C.c_int]
        L.ucg_io_dump_load.argtypes = [cp, ll, ll, C.c_int, _dp, cp, C.c_int]
        L.ucg_io_read_dump.argtypes = [cp, ll, cp, cp, ap, llp, cp, C.c_int]
        L.ucg_io_write_data.argtypes = [cp, ll, cp, ap, cp, C.c_int]
        L.ucg_io_data_header.argtypes = [cp, llp, _ip, _dp, _dp, _ip, cp, C.c_int]
        L.ucg_io_read_data.argtypes = [cp, ap, cp, C.c_int]
        L.ucg_io_write_restart.argtypes = [cp, ll, ap, cp, C.c_int]
        L.ucg_io_restart_header.argtypes = [cp, llp, llp, _ip, _dp, _dp, cp, C.c_int]
        L.ucg_io_read_restart.argtypes = [cp, ap, llp, cp, C.c_int]
        _SETUP = True
    return L


def _chk(rc, err):
    if rc:
        raise capi.UcgError(rc, err.value.decode())


def _errbuf():
    return C.create_string_buffer(512)


def _pack(atoms: dict, writable=False):
    """dict of arrays -> (IoAtoms, keep-alive list).  'tag' is accepted for 'id'."""
    A = IoAtoms()
    keep = {}
    src = dict(atoms)
    if "id" not in src and "tag" in src:
        src["id"] = src["tag"]
    n = None
    for k in INT_FIELDS + DBL_FIELDS:
        v = src.get(k)
        if v is None:
            continue
        dt = np.int32 if k in INT_FIELDS else np.float64
        arr = np.ascontiguousarray(v, dtype=dt)
        if writable and arr is v:
            arr = arr.copy()  # never write into the caller's arrays
        keep[k] = arr
        setattr(A, k, arr.ctypes.data_as(_ip if k in INT_FIELDS else _dp))
        if k != "mass":
            rows = arr.shape[0]
            if n is None:
                n = rows
            elif rows < n:
                raise ValueError(f"array {k} has {rows} rows, expected {n}")
    A.n = int(src.get("n", src.get("nlocal", n or 0)))
    A.ntypes = int(src.get("ntypes", (len(keep["mass"]) - 1) if "mass" in keep else 1))
    for d in range(3):
        A.boxlo[d] = float(src["boxlo"][d])
        A.boxhi[d] = float(src["boxhi"][d])
    return A, keep


def _unpack(A: IoAtoms, keep: dict) -> dict:
    n = A.n
    out = {k: (v if k == "mass" else v[:n]) for k, v in keep.items()}
    out["n"] = n
    out["ntypes"] = A.ntypes
    out["boxlo"] = np.array(A.boxlo[:])
    out["boxhi"] = np.array(A.boxhi[:])
    return out


def _alloc(n, ntypes, fields):
    out = {}
    for k in fields:
        if k == "mass":
            out[k] = np.zeros(ntypes + 1)
        elif k in INT_FIELDS:
            out[k] = np.zeros((n, _WIDTH[k]) if k in _WIDTH else n, np.int32)
        else:
            out[k] = np.zeros((n, _WIDTH[k]) if k in _WIDTH else n)
    out["boxlo"], out["boxhi"], out["ntypes"], out["n"] = np.zeros(3), np.zeros(3), ntypes, n
    return out


# ------------------------------------------------------------------ dump / read_dump

def write_dump(path, atoms, columns, timestep=0, modify=(), append=False):
    """one snapshot of `dump custom` (+ `dump_modify` keyword groups in `modify`); returns the atoms written"""
    L, (A, _keep), err = _lib(), _pack(atoms), _errbuf()
    nw = C.c_longlong(0)
    cols = columns if isinstance(columns, str) else " ".join(columns)
    mod = modify if isinstance(modify, str) else "\n".join(modify)
    _chk(L.ucg_io_dump_write(str(path).encode(), int(append), int(timestep), C.byref(A), cols.encode(),
                             mod.encode(), C.byref(nw), err, len(err)), err)
    return nw.value


def dump_snapshots(path):
    """[(timestep, natoms), ...] of a native text dump file"""
    L, err = _lib(), _errbuf()
    n = C.c_int(0)
    _chk(L.ucg_io_dump_scan(str(path).encode(), None, None, 0, C.byref(n), err, len(err)), err)
    ts, na = np.zeros(max(n.value, 1), np.int64), np.zeros(max(n.value, 1), np.int64)
    llp = C.POINTER(C.c_longlong)
    _chk(L.ucg_io_dump_scan(str(path).encode(), ts.ctypes.data_as(llp), na.ctypes.data_as(llp), n.value, C.byref(n),
                            err, len(err)), err)
    return [(int(ts[i]), int(na[i])) for i in range(n.value)]


def load_dump(path, timestep=-1):
    """every column of one snapshot: dict(timestep, natoms, boxlo, boxhi, columns=[names], values[natoms, ncol])"""
    L, err = _lib(), _errbuf()
    ft, na = C.c_longlong(0), C.c_longlong(0)
    lo, hi = np.zeros(3), np.zeros(3)
    cols = C.create_string_buffer(4096)
    _chk(L.ucg_io_dump_header(str(path).encode(), int(timestep), C.byref(ft), C.byref(na), lo.ctypes.data_as(_dp),
                              hi.ctypes.data_as(_dp), cols, len(cols), err, len(err)), err)
    names = cols.value.decode().split()
    vals = np.zeros((na.value, len(names)))
    _chk(L.ucg_io_dump_load(str(path).encode(), ft.value, na.value, len(names), vals.ctypes.data_as(_dp), err,
                            len(err)), err)
    return dict(timestep=ft.value, natoms=na.value, boxlo=lo, boxhi=hi, columns=names, values=vals)


def read_dump(path, timestep, fields, atoms, options=""):
    """`read_dump path timestep fields options` on the arrays of `atoms` (matched by ID).  Returns (atoms', stats)
    with stats = dict(snapshot=, replaced=, trimmed=, natoms=)."""
    L, err = _lib(), _errbuf()
    A, keep = _pack(atoms, writable=True)
    st = (C.c_longlong * 4)()
    f = fields if isinstance(fields, str) else " ".join(fields)
    _chk(L.ucg_io_read_dump(str(path).encode(), int(timestep), f.encode(), options.encode(), C.byref(A), st, err,
                            len(err)), err)
    out = dict(atoms)
    out.update(_unpack(A, keep))
    if "tag" in atoms and "id" not in atoms:
        out["tag"] = out.pop("id")
    return out, dict(snapshot=st[0], replaced=st[1], trimmed=st[2], natoms=st[3])


# ------------------------------------------------------------------ data file

def write_data(path, atoms, timestep=0, units="lj"):
    L, (A, _keep), err = _lib(), _pack(atoms), _errbuf()
    _chk(L.ucg_io_write_data(str(path).encode(), int(timestep), units.encode(), C.byref(A), err, len(err)), err)


def read_data(path):
    """-> dict of arrays in file order (id, molecule, type, q, x, image, ucgstate, ucgl, ucgml, ucgp = -1, v, ucgvl,
    mass) + boxlo / boxhi / ntypes / n"""
    L, err = _lib(), _errbuf()
    na, nt, hv = C.c_longlong(0), C.c_int(0), C.c_int(0)
    lo, hi = np.zeros(3), np.zeros(3)
    _chk(L.ucg_io_data_header(str(path).encode(), C.byref(na), C.byref(nt), lo.ctypes.data_as(_dp),
                              hi.ctypes.data_as(_dp), C.byref(hv), err, len(err)), err)
    arrays = _alloc(na.value, nt.value, ("id", "molecule", "type", "q", "x", "image", "ucgstate", "ucgl", "ucgml",
                                         "ucgp", "v", "ucgvl", "mass"))
    A, keep = _pack(arrays, writable=True)
    _chk(L.ucg_io_read_data(str(path).encode(), C.byref(A), err, len(err)), err)
    return _unpack(A, keep)


# ------------------------------------------------------------------ restart container

def write_restart(path, atoms, timestep=0):
    L, (A, _keep), err = _lib(), _pack(atoms), _errbuf()
    _chk(L.ucg_io_write_restart(str(path).encode(), int(timestep), C.byref(A), err, len(err)), err)


def read_restart(path):
    L, err = _lib(), _errbuf()
    ts, na, nt = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
    lo, hi = np.zeros(3), np.zeros(3)
    _chk(L.ucg_io_restart_header(str(path).encode(), C.byref(ts), C.byref(na), C.byref(nt), lo.ctypes.data_as(_dp),
                                 hi.ctypes.data_as(_dp), err, len(err)), err)
    arrays = _alloc(na.value, nt.value, ("id", "type", "x", "molecule", "v", "q", "image", "ucgstate", "ucgl", "ucgml",
                                         "ucgvl", "ucgp", "mass"))
    A, keep = _pack(arrays, writable=True)
    _chk(L.ucg_io_read_restart(str(path).encode(), C.byref(A), C.byref(ts), err, len(err)), err)
    out = _unpack(A, keep)
    out["timestep"] = ts.value
    return out


# ------------------------------------------------------------------ glue to the resident loop

def atoms_of(ctx, boxlo, boxhi, ntypes=2, mass=None, molecule=None):
    """Context.atoms_download() + box, in the shape the writers take"""
    a = ctx.atoms_download()
    n = a["nlocal"]
    out = {k: a[k][:n] for k in ("x", "v", "f", "type", "ucgstate", "ucgl", "ucgvl", "ucgml", "ucgp", "ucgforce")}
    out["id"] = a["tag"][:n]
    out.update(boxlo=np.asarray(boxlo, float), boxhi=np.asarray(boxhi, float), ntypes=ntypes, n=n)
    if mass is not None:
        out["mass"] = np.asarray(mass, float)
    if molecule is not None:
        out["molecule"] = molecule
    return out
