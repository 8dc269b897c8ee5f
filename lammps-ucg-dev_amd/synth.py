"""Synthetic UCG inputs: LJ-like table files, state-settings files, bead lattices.

The reference ships no input decks, table files or data files, so
every run -- tests, bench.py, smoke() -- generates its inputs here, in the on-disk
formats the reference's parsers accept:

* table files in the upstream ``pair_style table`` format parsed by
  ``read_table``/``param_extract`` (UCG/pair_table_ucgld.cpp:897-1017, 1067-1102);
* state-settings files parsed by ``read_state_settings``
  (UCG/pair_table_ucgld.cpp:565-652; density variant
  UCG/pair_table_ucg_bethe_density.cpp:778-893);
* per-bead arrays in the column order of the ``atom_style ucg`` data file
  (``id mol type q x y z ucgstate ucgl ucgml`` / ``id vx vy vz ucgvl``,
  UCG/atom_vec_ucg.cpp:87-90) with the ``data_atom_post`` clamps applied
  (UCG/atom_vec_ucg.cpp:145-170).

Defaults follow BASELINE.md: LJ reduced units, rho* = 0.8, rc = 2.5, skin = 0.3,
eps_00 = 1.0, eps_01 = eps_10 = 0.8, eps_11 = 0.5, sigma = 1, file grid ``N 2000 R 0.6 2.5``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

DEFAULT_EPS = {"00": 1.0, "01": 0.8, "10": 0.8, "11": 0.5}


def lj_energy_force(r: np.ndarray, eps: float, sigma: float = 1.0, extra_exp: float = 0.0):
    """u(r) = 4 eps [(s/r)^12 - (s/r)^6] (+ extra_exp * exp(-r)); returns (u, -du/dr)."""
    sr6 = (sigma / r) ** 6
    u = 4.0 * eps * (sr6 * sr6 - sr6)
    f = 24.0 * eps * (2.0 * sr6 * sr6 - sr6) / r
    if extra_exp != 0.0:
        u = u + extra_exp * np.exp(-r)
        f = f + extra_exp * np.exp(-r)
    return u, f


def bitmap_r(n: int, rlo: float, rhi: float) -> np.ndarray:
    """r of the n = 2^bits entries of a BITMAP table section: entry i holds the float whose index bits are i
    (upstream LAMMPS Pair::init_bitmap + UCG/pair_table_ucgld.cpp:954-972)"""
    bits = int(n).bit_length() - 1
    if 1 << bits != n:
        raise ValueError("a BITMAP section holds 2^bits entries")
    nlowermin = int(np.floor(np.log2(rlo * rlo)))
    nexpbits, available = 0, 2.0
    while available < rhi * rhi / 2.0 ** nlowermin:
        nexpbits += 1
        available = 2.0 ** (2.0 ** nexpbits)
    nshift = 24 - (bits - nexpbits + 1)
    nmask = (1 << (bits + nshift)) - 1
    as_int = lambda v: int(np.array([v], np.float32).view(np.int32)[0])  # noqa: E731
    masklo, maskhi = as_int(rlo * rlo) & ~nmask, as_int(rhi * rhi) & ~nmask
    i = np.arange(n, dtype=np.int64) << nshift
    lo = (i | masklo).astype(np.int32).view(np.float32)
    hi = (i | maskhi).astype(np.int32).view(np.float32)
    rsq = np.where(lo.astype(np.float64) < rlo * rlo, hi, lo)
    return np.sqrt(rsq.astype(np.float32)).astype(np.float64)


def write_table_file(path: str, sections: dict, n: int = 2000, rlo: float = 0.6, rhi: float = 2.5,
                     rmode: str = "R") -> str:
    """Write one table file holding several keyword sections.

    ``sections`` maps keyword -> (eps, extra_exp).  ``rmode`` is "R" (uniform in r) or
    "RSQ" (uniform in r^2), written on the parameter line as the reference expects.
    """
    rlo0 = rlo
    with open(path, "w") as fh:
        fh.write("# UCG synthetic LJ-like tables (generated)\n\n")
        for key, spec in sections.items():
            eps, extra = spec[0], spec[1]
            rlo = spec[2] if len(spec) > 2 and spec[2] is not None else rlo0
            if rmode == "R":
                r = rlo + (rhi - rlo) * np.arange(n) / (n - 1)
            elif rmode == "RSQ":
                r = np.sqrt(rlo * rlo + (rhi * rhi - rlo * rlo) * np.arange(n) / (n - 1))
            elif rmode == "BITMAP":
                r = bitmap_r(n, rlo, rhi)
            else:
                raise ValueError(rmode)
            u, f = lj_energy_force(r, eps, 1.0, extra)
            fh.write(f"{key}\n")
            fh.write(f"N {n} {rmode} {float(rlo)!r} {float(rhi)!r}\n\n")
            for i in range(n):
                fh.write(f"{i + 1} {float(r[i])!r} {float(u[i])!r} {float(f[i])!r}\n")
            fh.write("\n")
    return path


def write_state_settings(path: str, mu=(0.0, 0.5), density=None, entropy: bool = False) -> str:
    """One actual type with two formal types (1, 2).

    Plain form (table_ucgld / table_ucg_bethe)::

        1 2 2
        1 2
        1 2
        mu0 mu1

    Density form (table_ucg_bethe_density), ``density=(rho_th, r_th)``::

        1 2 2
        1 2
        1 2 density no_entropy
        rho_th r_th
        mu0 mu1
    """
    with open(path, "w") as fh:
        fh.write("1 2 2\n")
        fh.write("1 2\n")
        if density is None:
            fh.write("1 2\n")
        else:
            fh.write(f"1 2 density {'entropy' if entropy else 'no_entropy'} \n")
            fh.write(f"{float(density[0])!r} {float(density[1])!r}\n")
        fh.write(f"{float(mu[0])!r} {float(mu[1])!r}\n")
    return path


@dataclass
class Deck:
    """The handful of files + command arguments that define one pair-style setup."""
    workdir: str
    table_file: str
    conf_file: str
    tabstyle: str = "spline"
    tablength: int = 1024
    cut: float = 2.5
    extra_keywords: tuple = ()

    def pair_style_args(self):
        return [self.tabstyle, str(self.tablength), self.conf_file, *self.extra_keywords]

    def pair_coeff_args(self):
        # pair_coeff 1 1 2 2  (file keyword cutoff) x 4, order 00 01 10 11
        args = ["1", "1", "2", "2"]
        for key in ("UCG_00", "UCG_01", "UCG_10", "UCG_11"):
            args += [self.table_file, key, repr(float(self.cut))]
        return args


def make_deck(workdir: str, tabstyle: str = "spline", tablength: int = 1024, mu=(0.0, 0.5),
              density=None, entropy: bool = False, eps=None, extra11: float = 0.0, n_file: int = 2000,
              rlo: float = 0.6, rhi: float = 2.5, cut: float = 2.5, rmode: str = "R",
              extra_keywords=(), rlo11=None) -> Deck:
    os.makedirs(workdir, exist_ok=True)
    eps = dict(DEFAULT_EPS if eps is None else eps)
    sections = {
        "UCG_00": (eps["00"], 0.0),
        "UCG_01": (eps["01"], 0.0),
        "UCG_10": (eps["10"], 0.0),
        "UCG_11": (eps["11"], extra11, rlo11),
    }
    tfile = write_table_file(os.path.join(workdir, "ucg_lj.table"), sections, n_file, rlo, rhi, rmode)
    cfile = write_state_settings(os.path.join(workdir, "ucg.conf"), mu, density, entropy)
    return Deck(workdir, tfile, cfile, tabstyle, tablength, cut, tuple(extra_keywords))


@dataclass
class MultiDeck:
    """A deck with several ACTUAL atom types, each with two formal types: actual type a (1-based) has the
    formal types (2a-1, 2a); atom->ntypes = 2 * n_actual.  One pair_coeff command per actual pair i <= j."""
    workdir: str
    table_file: str
    conf_file: str
    n_actual: int
    tabstyle: str = "spline"
    tablength: int = 1024
    cut: float = 2.5
    extra_keywords: tuple = ()

    @property
    def ntypes(self):
        return 2 * self.n_actual

    def pair_style_args(self):
        return [self.tabstyle, str(self.tablength), self.conf_file, *self.extra_keywords]

    def pair_coeff_commands(self):
        """argument lists of `pair_coeff i j Ns_i Ns_j (file keyword cutoff) x 4` (UCG/pair_table_ucgld.cpp:719-745)"""
        cmds = []
        for i in range(1, self.n_actual + 1):
            for j in range(i, self.n_actual + 1):
                args = [str(i), str(j), "2", "2"]
                for a in (0, 1):
                    for b in (0, 1):
                        args += [self.table_file, f"P{i}{j}_{a}{b}", repr(float(self.cut))]
                cmds.append(args)
        return cmds


def make_multi_deck(workdir: str, n_actual: int = 2, tabstyle: str = "spline", tablength: int = 1024,
                    mu=(0.0, 0.5), n_file: int = 1200, rlo: float = 0.6, rhi: float = 2.5, cut: float = 2.5,
                    extra_keywords=(), density=None, entropy: bool = False, extra11: float = 0.0) -> MultiDeck:
    """n_actual two-state atom types; the LJ-like well depths of DEFAULT_EPS are scaled per actual pair so that
    every one of the 4 * n_actual (n_actual + 1) / 2 tables is different (u_ab of pair (i, j), i < j, is NOT
    symmetric in the states: state a belongs to type i, state b to type j).  ``density=(rho_th, r_th)`` writes the
    settings file of table_ucg_bethe_density (every type a density type; ``extra11`` adds extra11 * exp(-r) to the
    (1,1) tables so that J never vanishes)."""
    os.makedirs(workdir, exist_ok=True)
    sections = {}
    for i in range(1, n_actual + 1):
        for j in range(i, n_actual + 1):
            scale = 1.0 - 0.12 * (j - i) - 0.05 * (i + j - 2)
            for a in (0, 1):
                for b in (0, 1):
                    eps = DEFAULT_EPS[f"{a}{b}"] * scale * (1.0 + (0.07 * (a - b) if i != j else 0.0))
                    sections[f"P{i}{j}_{a}{b}"] = (eps, extra11 if (a == 1 and b == 1) else 0.0)
    tfile = write_table_file(os.path.join(workdir, "ucg_multi.table"), sections, n_file, rlo, rhi, "R")
    cfile = os.path.join(workdir, "ucg_multi.conf")
    with open(cfile, "w") as fh:
        fh.write(f"{n_actual} {2 * n_actual} 2\n")
        for a in range(1, n_actual + 1):
            fh.write(f"{a} 2\n")
            if density is None:
                fh.write(f"{2 * a - 1} {2 * a}\n")
            else:
                fh.write(f"{2 * a - 1} {2 * a} density {'entropy' if entropy else 'no_entropy'} \n")
                fh.write(f"{float(density[0])!r} {float(density[1])!r}\n")
            fh.write(f"{float(mu[0] + 0.1 * (a - 1))!r} {float(mu[1] - 0.05 * (a - 1))!r}\n")
    return MultiDeck(workdir, tfile, cfile, n_actual, tabstyle, tablength, cut, tuple(extra_keywords))


def write_cluster_switch_files(workdir: str, prob_on: float, types_on, types_off, contacts):
    """rates file (probON / nSwitchTypes / ON types / OFF types) and contact-map file
    (label nContactTypes / label nAtomsPerContact / one type pair per line), UCG/fix_cluster_switch.cpp:207-344"""
    os.makedirs(workdir, exist_ok=True)
    rates = os.path.join(workdir, "rates.txt")
    with open(rates, "w") as fh:
        fh.write(f"{float(prob_on)!r}\n{len(types_on)}\n")
        fh.write(" ".join(str(int(t)) for t in types_on) + "\n")
        fh.write(" ".join(str(int(t)) for t in types_off) + "\n")
    cfile = os.path.join(workdir, "contacts.txt")
    with open(cfile, "w") as fh:
        fh.write(f"nContactTypes {len(contacts)}\nnAtomsPerContact 1\n")
        for a, b in contacts:
            fh.write(f"{int(a)} {int(b)}\n")
    return rates, cfile


@dataclass
class Beads:
    """Per-bead arrays in the layout LAMMPS hands to a pair style (AoS x[n][3])."""
    n: int
    boxlo: np.ndarray
    boxhi: np.ndarray
    x: np.ndarray        # [n,3] f64
    v: np.ndarray        # [n,3] f64
    type: np.ndarray     # [n] i32 (actual type, 1-based)
    tag: np.ndarray      # [n] i32 (1-based id)
    mask: np.ndarray     # [n] i32
    ucgstate: np.ndarray  # [n] i32
    ucgl: np.ndarray     # [n] f64
    ucgvl: np.ndarray    # [n] f64
    ucgml: np.ndarray    # [n] f64
    ucgp: np.ndarray     # [n] f64 (-1 = unassigned, data_atom_post)
    mass: np.ndarray = field(default_factory=lambda: np.array([0.0, 1.0, 1.0]))  # per type, 1-based
    ntypes: int = 2
    molecule: np.ndarray = None  # [n] i32 molecule id (atom->molecule), optional


def make_beads(ncell: int, rho: float = 0.8, jitter: float = 0.1, temp: float = 1.0, seed: int = 12345,
               lattice: str = "sc", ucgml: float = 10.0) -> Beads:
    """ncell^3 simple-cubic (or 4 ncell^3 fcc) beads at reduced density rho, jittered by
    U(-jitter a, jitter a); Gaussian velocities at T* = temp with zero total momentum."""
    rng = np.random.default_rng(seed)
    if lattice == "sc":
        a = (1.0 / rho) ** (1.0 / 3.0)
        g = np.arange(ncell, dtype=np.float64)
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1) * a + 0.5 * a
    elif lattice == "fcc":
        a = (4.0 / rho) ** (1.0 / 3.0)
        g = np.arange(ncell, dtype=np.float64)
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        base = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        offs = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
        x = (base[:, None, :] + offs[None, :, :]).reshape(-1, 3) * a + 0.25 * a
    else:
        raise ValueError(lattice)
    n = x.shape[0]
    L = ncell * a
    x = x + rng.uniform(-jitter * a, jitter * a, size=x.shape)
    x = np.mod(x, L)
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = rng.normal(0.0, np.sqrt(temp), size=(n, 3))
    v -= v.mean(axis=0, keepdims=True)
    ucgstate = (rng.uniform(size=n) < 0.5).astype(np.int32)
    ucgl = rng.uniform(size=n)
    # data_atom_post clamps (UCG/atom_vec_ucg.cpp:155-165)
    ucgl = np.clip(ucgl, 0.0, 1.0)
    ucgstate = np.clip(ucgstate, 0, 1).astype(np.int32)
    return Beads(
        n=n,
        boxlo=np.zeros(3),
        boxhi=np.full(3, L),
        x=x,
        v=np.ascontiguousarray(v),
        type=np.ones(n, dtype=np.int32),
        tag=np.arange(1, n + 1, dtype=np.int32),
        mask=np.ones(n, dtype=np.int32),
        ucgstate=ucgstate,
        ucgl=np.ascontiguousarray(ucgl),
        ucgvl=np.zeros(n),
        ucgml=np.full(n, float(ucgml)),
        ucgp=np.full(n, -1.0),
    )


def make_cluster(n: int, box: float = 40.0, radius: float = 3.0, seed: int = 7, min_dist: float = 0.85) -> Beads:
    """Open-boundary droplet in a box much larger than the cluster (no bead has a
    periodic image within the ghost cutoff): the geometry on which the reference's
    table_ucg_bethe_density arithmetic is well defined."""
    rng = np.random.default_rng(seed)
    pts = []
    c = np.full(3, 0.5 * box)
    while len(pts) < n:
        p = c + rng.uniform(-radius, radius, size=3)
        if np.linalg.norm(p - c) > radius:
            continue
        if pts and np.min(np.linalg.norm(np.array(pts) - p, axis=1)) < min_dist:
            continue
        pts.append(p)
    x = np.ascontiguousarray(np.array(pts))
    b = make_beads(2, seed=seed)
    ucgstate = (rng.uniform(size=n) < 0.5).astype(np.int32)
    return Beads(
        n=n, boxlo=np.zeros(3), boxhi=np.full(3, box), x=x, v=np.zeros((n, 3)),
        type=np.ones(n, dtype=np.int32), tag=np.arange(1, n + 1, dtype=np.int32),
        mask=np.ones(n, dtype=np.int32), ucgstate=ucgstate, ucgl=rng.uniform(size=n),
        ucgvl=np.zeros(n), ucgml=np.full(n, 10.0), ucgp=np.full(n, -1.0), mass=b.mass,
    )
