"""TEST INFRASTRUCTURE ONLY (see oracle/README in orc.h): pure-Python restatement of the reference's on-disk
formats for atom style ucg, used by tests/test_io.py to check csrc/ucg_io.cpp byte for byte.

  dump_text()   DumpCustom::header_item (dump_custom.cpp:650-670), default column formats "%d" / "%g"
                (:142-152), one snprintf per column and a newline (:1395-1417), keywords ucgstate (INT) / ucgl / ucgp
                (DOUBLE) (:1672-1688, pack :3552-3578), thresholds ANDed (:1182-1209, :2150-2155), sort by id
  read_dump()   ReadDump::process_atoms (read_dump.cpp:823-942): match by ID, overwrite the listed fields,
                images from ix/iy/iz (zero when not wrapped), trim by copying the last atom into the hole
  data_post()   AtomVecUCG::data_atom_post (UCG/atom_vec_ucg.cpp:145-170)

Python's % operator formats through the same C printf conversions for %d / %g / %e.
"""
import numpy as np

INT_COLS = {"id", "mol", "type", "ix", "iy", "iz", "ucgstate"}


def column(atoms, name, i):
    lo, hi = atoms["boxlo"], atoms["boxhi"]
    img = atoms.get("image")
    ax = {"x": 0, "y": 1, "z": 2}
    if name == "id":
        return atoms["id"][i]
    if name == "mol":
        return atoms["molecule"][i]
    if name == "type":
        return atoms["type"][i]
    if name == "mass":
        return atoms["mass"][atoms["type"][i]]
    if name in ax:
        return atoms["x"][i][ax[name]]
    if name in ("xs", "ys", "zs"):
        d = ax[name[0]]
        return (atoms["x"][i][d] - lo[d]) * (1.0 / (hi[d] - lo[d]))
    if name in ("xu", "yu", "zu"):
        d = ax[name[0]]
        return atoms["x"][i][d] + (img[i][d] if img is not None else 0) * (hi[d] - lo[d])
    if name in ("ix", "iy", "iz"):
        return img[i][ax[name[1]]] if img is not None else 0
    if name in ("vx", "vy", "vz"):
        return atoms["v"][i][ax[name[1]]]
    if name in ("fx", "fy", "fz"):
        return atoms["f"][i][ax[name[1]]]
    if name == "q":
        return atoms["q"][i] if atoms.get("q") is not None else 0.0
    return atoms[name][i]  # ucgstate ucgl ucgp ucgvl ucgml ucgforce


OPS = {"<": lambda a, b: a < b, "<=": lambda a, b: a <= b, ">": lambda a, b: a > b, ">=": lambda a, b: a >= b,
       "==": lambda a, b: a == b, "!=": lambda a, b: a != b}


def dump_text(atoms, columns, timestep=0, thresh=(), sort_id=False, fmt_float=None, fmt_int=None, boundary="pp pp pp"):
    cols = columns.split()
    n = len(atoms["id"])
    chosen = [i for i in range(n) if all(OPS[op](float(column(atoms, a, i)), float(v)) for a, op, v in thresh)]
    if sort_id:
        chosen.sort(key=lambda i: atoms["id"][i])
    out = ["ITEM: TIMESTEP\n%d\nITEM: NUMBER OF ATOMS\n%d\n" % (timestep, len(chosen)), "ITEM: BOX BOUNDS %s\n" % boundary]
    for d in range(3):
        out.append("%1.16e %1.16e\n" % (atoms["boxlo"][d], atoms["boxhi"][d]))
    out.append("ITEM: ATOMS %s\n" % " ".join(cols))
    for i in chosen:
        words = []
        for c in cols:
            v = column(atoms, c, i)
            if c in INT_COLS:
                words.append((fmt_int or "%d") % int(v))
            else:
                words.append((fmt_float or "%g") % float(v))
        out.append(" ".join(words) + "\n")
    return "".join(out)


def parse_dump(text):
    """-> list of snapshots dict(timestep, natoms, boxlo, boxhi, columns, values)"""
    lines = text.split("\n")
    snaps, k = [], 0
    while k < len(lines):
        if not lines[k].startswith("ITEM: TIMESTEP"):
            k += 1
            continue
        ts = int(lines[k + 1])
        n = int(lines[k + 3])
        lo, hi = np.zeros(3), np.zeros(3)
        for d in range(3):
            lo[d], hi[d] = (float(w) for w in lines[k + 5 + d].split())
        cols = lines[k + 8][len("ITEM: ATOMS "):].split()
        vals = np.array([[float(w) for w in lines[k + 9 + i].split()] for i in range(n)]).reshape(n, len(cols))
        snaps.append(dict(timestep=ts, natoms=n, boxlo=lo, boxhi=hi, columns=cols, values=vals))
        k += 9 + n
    return snaps


def read_dump(snap, fields, atoms, box=True, replace=True, trim=False, wrapped=True):
    """atoms: dict of arrays (copied); returns (atoms', stats)"""
    A = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in atoms.items()}
    cols = snap["columns"]
    lo, hi = (snap["boxlo"], snap["boxhi"]) if box else (A["boxlo"], A["boxhi"])
    where = {int(t): i for i, t in enumerate(A["id"])}
    n = len(A["id"])
    updated = np.zeros(n, bool)
    nreplace = 0
    ax = {"x": 0, "y": 1, "z": 2}
    for row in snap["values"]:
        m = where.get(int(row[cols.index("id")]))
        if m is None:
            continue
        updated[m] = True
        if not replace:
            continue
        nreplace += 1
        img = [int(A["image"][m][d]) for d in range(3)] if A.get("image") is not None else [0, 0, 0]
        for f in fields.split():
            if f in ax:
                d = ax[f]
                if f in cols:
                    A["x"][m][d] = row[cols.index(f)]
                elif f + "s" in cols:
                    A["x"][m][d] = row[cols.index(f + "s")] * (hi[d] - lo[d]) + lo[d]
                else:
                    A["x"][m][d] = row[cols.index(f + "u")]
            elif f in ("vx", "vy", "vz"):
                A["v"][m][ax[f[1]]] = row[cols.index(f)]
            elif f in ("fx", "fy", "fz"):
                A["f"][m][ax[f[1]]] = row[cols.index(f)]
            elif f in ("ix", "iy", "iz"):
                img[ax[f[1]]] = int(row[cols.index(f)])
            elif f == "ucgstate":
                A["ucgstate"][m] = int(row[cols.index(f)])
            elif f in ("ucgl", "ucgp", "q"):
                A[f][m] = row[cols.index(f)]
        if not wrapped:
            img = [0, 0, 0]
        if A.get("image") is not None:
            A["image"][m] = img
    if box:
        A["boxlo"], A["boxhi"] = np.array(lo), np.array(hi)
    ntrim = 0
    if trim:
        per_atom = [k for k, v in A.items() if isinstance(v, np.ndarray) and k not in ("boxlo", "boxhi", "mass")]
        nlocal, i = n, 0
        upd = updated.copy()
        while i < nlocal:
            if not upd[i]:
                for k in per_atom:
                    A[k][i] = A[k][nlocal - 1]
                upd[i] = upd[nlocal - 1]
                nlocal -= 1
                ntrim += 1
            else:
                i += 1
        for k in per_atom:
            A[k] = A[k][:nlocal]
    return A, dict(snapshot=snap["natoms"], replaced=nreplace, trimmed=ntrim, natoms=len(A["id"]))


def data_post(ucgstate, ucgl):
    """clamp lambda to [0,1], the state to {0,1}; ucgp = -1 (unassigned)"""
    lam = np.where(ucgl < 0, 0.0, np.where(ucgl > 1, 1.0, ucgl))
    st = np.clip(ucgstate, 0, 1)
    return st.astype(np.int32), lam, np.full(len(lam), -1.0)
