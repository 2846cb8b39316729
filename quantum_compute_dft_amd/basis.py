"""Molecule + Gaussian basis -> shell table for DFT_EvalAO.

Host-side counterpart of what the reference gets from PySCF's `gto.Mole`
(grid.py:23-31,42-47: atom text, `basis`, `mol.build()`), restricted to what the
AO-on-grid kernel needs: shell centres (bohr), angular momenta, primitive
exponents and *normalised* contraction coefficients, AO column offsets in
PySCF's spherical ordering (atoms in input order; per atom shells by l; p as
x,y,z; d,f as m=-l..l).

Normalisation follows PySCF/libcint: primitives are radially normalised
(gto_norm), the contraction is then renormalised to unit self-overlap
(_nomalize_contracted_ao); the angular normalisation lives in the kernel's
real solid harmonics.

Basis data: STO-3G (H, C, N, O, P, S) from the Hehre-Stewart-Pople scale
factors and def2-SVP (H, C, N, O) as published by Weigend & Ahlrichs (2005),
typed in from memory -- there is no network or PySCF in the build image to
fetch or diff them ("unpinned inputs").  Other elements /
bases can be supplied in the same dict format via `register_basis`.
"""
import math
import os
from dataclasses import dataclass

import numpy as np

BOHR = 0.52917721092  # Angstrom per bohr, the constant PySCF uses (CODATA 2010)

ELEMENTS = ["X", "H", "He", "Li", "Be", "B", "C", "N", "O", "F", "Ne", "Na", "Mg", "Al", "Si",
            "P", "S", "Cl", "Ar"]
_Z = {s.upper(): i for i, s in enumerate(ELEMENTS)}

# STO-3G least-squares fits to Slater 1s / 2sp / 3sp functions of unit exponent
_STO3G_1S = ((2.227660584, 0.405771156, 0.109818),
             (0.15432897, 0.53532814, 0.44463454))
_STO3G_2SP = ((0.994203, 0.231031, 0.0751386),
              (-0.09996723, 0.39951283, 0.70011547), (0.15591627, 0.60768372, 0.39195739))
_STO3G_3SP = ((0.4828540806, 0.1347150629, 0.0527268058),
              (-0.2196203690, 0.2255954336, 0.9003984260),
              (0.01058760429, 0.5951670053, 0.4620010120))

# tabulated STO-3G exponents (EMSL); coefficients are the universal ones above
_STO3G_EXPS = {
    "H": [(3.42525091, 0.62391373, 0.16885540)],
    "C": [(71.6168370, 13.0450960, 3.5305122), (2.9412494, 0.6834831, 0.2222899)],
    "N": [(99.1061690, 18.0523120, 4.8856602), (3.7804559, 0.8784966, 0.2857144)],
    "O": [(130.7093200, 23.8088610, 6.4436083), (5.0331513, 1.1695961, 0.3803890)],
    "P": [(468.3656378, 85.31338559, 23.08913156), (28.03263958, 6.514182577, 2.118614352),
          (1.743103231, 0.4863213771, 0.1903428909)],
    "S": [(533.1257359, 97.10951830, 26.28162542), (33.32975173, 7.745117521, 2.518952599),
          (2.029194274, 0.5661400518, 0.2215833792)],
}


def _sto3g(sym):
    shells = []
    ex = _STO3G_EXPS[sym]
    shells.append((0, list(zip(ex[0], _STO3G_1S[1]))))
    if len(ex) > 1:
        shells.append((0, list(zip(ex[1], _STO3G_2SP[1]))))
        shells.append((1, list(zip(ex[1], _STO3G_2SP[2]))))
    if len(ex) > 2:
        shells.append((0, list(zip(ex[2], _STO3G_3SP[1]))))
        shells.append((1, list(zip(ex[2], _STO3G_3SP[2]))))
    return shells


# (l, [(exponent, coefficient), ...]) per shell
_DEF2_SVP = {
    "H": [
        (0, [(13.0107010, 0.19682158e-1), (1.9622572, 0.13796524), (0.44453796, 0.47831935)]),
        (0, [(0.12194962, 1.0)]),
        (1, [(0.8000000, 1.0)]),
    ],
    "C": [
        (0, [(1238.4016938, 0.0054568832082), (186.29004992, 0.040638409211),
             (42.251176346, 0.18025593888), (11.676557932, 0.46315121755),
             (3.5930506482, 0.44087173314)]),
        (0, [(0.40245147363, 1.0)]),
        (0, [(0.13090182668, 1.0)]),
        (1, [(9.4680970621, 0.038387871728), (2.0103545142, 0.21117025112),
             (0.54771004707, 0.51328172114)]),
        (1, [(0.15268613795, 1.0)]),
        (2, [(0.8000000, 1.0)]),
    ],
    "N": [
        (0, [(1712.8415853, -0.0053934125305), (257.64812677, -0.040221581118),
             (58.458245853, -0.17931144990), (16.198367905, -0.46376317823),
             (5.0052600809, -0.44171422662)]),
        (0, [(0.58731856571, 1.0)]),
        (0, [(0.18764592253, 1.0)]),
        (1, [(13.571470233, -0.040072398852), (2.9257372874, -0.21807045028),
             (0.79927750754, -0.51294466049)]),
        (1, [(0.21954348034, 1.0)]),
        (2, [(1.0000000, 1.0)]),
    ],
    "O": [
        (0, [(2266.1767785, -0.0053431809926), (340.87010191, -0.039890039230),
             (77.363135167, -0.17853911985), (21.479644940, -0.46427684959),
             (6.6589433124, -0.44309745172)]),
        (0, [(0.80975975668, 1.0)]),
        (0, [(0.25530772234, 1.0)]),
        (1, [(17.721504317, 0.043394573193), (3.8635505440, 0.23094120765),
             (1.0480920883, 0.51375311064)]),
        (1, [(0.27641544411, 1.0)]),
        (2, [(1.2000000, 1.0)]),
    ],
}

# def2-TZVP for H and C only (Benzene: 222, Anthracene: 494 functions -- BASELINE config 3), written
# down from memory like the tables above and NOT verifiable offline beyond the H-atom energy
# (-0.49981 Ha, tests/test_integrals.py): treat energies in this basis as "def2-TZVP-shaped".
_DEF2_TZVP = {
    "H": [
        (0, [(34.0613410, 0.60251978e-2), (5.1235746, 0.45021094e-1), (1.1646626, 0.20189726)]),
        (0, [(0.32723041, 1.0)]),
        (0, [(0.10307241, 1.0)]),
        (1, [(0.8000000, 1.0)]),
    ],
    "C": [
        (0, [(13575.349682, 0.22245814352e-3), (2035.2333680, 0.17232738252e-2),
             (463.22562359, 0.89255715314e-2), (131.20019598, 0.35727984502e-1),
             (42.853015891, 0.11076259931), (15.584185766, 0.24295627626)]),
        (0, [(6.2067138508, 0.41440263448), (2.5764896527, 0.23744968655)]),
        (0, [(0.57696339419, 1.0)]),
        (0, [(0.22972831358, 1.0)]),
        (0, [(0.95164440028e-1, 1.0)]),
        (1, [(34.697232244, 0.53333657805e-2), (7.9582622826, 0.35864109092e-1),
             (2.3780826883, 0.14215873329), (0.81433208183, 0.34270471845)]),
        (1, [(0.28887547253, 1.0)]),
        (1, [(0.10056823671, 1.0)]),
        (2, [(1.09700000, 1.0)]),
        (2, [(0.31800000, 1.0)]),
        (3, [(0.76100000, 1.0)]),
    ],
}

_BASIS_SETS = {
    "sto-3g": {s: _sto3g(s) for s in _STO3G_EXPS},
    "def2-svp": _DEF2_SVP,
    "def2-tzvp": _DEF2_TZVP,
}


def register_basis(name, table):
    """table: {element symbol: [(l, [(exp, coef), ...]), ...]}."""
    _BASIS_SETS[name.lower().replace("_", "-")] = table


def basis_names():
    return sorted(_BASIS_SETS)


_L_OF = {"S": 0, "P": 1, "D": 2, "F": 3}


def load_basis_file(path, name=None):
    """Register a basis from a text file in NWChem or Gaussian94 format (what the Basis Set Exchange
    exports; the reference takes the same data from PySCF's bundled copies, grid.py:45).  General
    contractions (several coefficient columns) become one shell per column, SP shells an s and a p shell.
    Returns the registered name.  Tables not shipped here -- def2-SVP for P and S, def2-TZVP beyond H and C:
    their 10-digit numbers are not reproducible from memory and nothing may be fetched in the build image --
    come in this way; `check_table` below applies the same invariants to them as to the shipped ones."""
    table, sym, block = {}, None, None          # block = (kind, rows) being read for element `sym`

    def flush():
        nonlocal block
        if block and sym and block[1]:
            kind, rows = block
            ncol = max(len(r) for r in rows) - 1
            for k in range(ncol):
                l = _L_OF[kind] if kind != "SP" else k      # SP: column 0 -> s, column 1 -> p
                prims = [(r[0], r[1 + k]) for r in rows if len(r) > 1 + k and r[1 + k] != 0.0]
                if prims:
                    table.setdefault(sym, []).append((l, prims))
        block = None

    num = lambda t: float(t.replace("D", "E").replace("d", "e"))
    for raw in open(path):
        line = raw.split("#")[0].split("!")[0].strip()
        if not line or line.upper().startswith(("BASIS", "END")):
            continue
        if line.startswith("****"):
            flush(); sym = None
            continue
        tok = line.split()
        if block is not None:
            try:
                block[1].append([num(t) for t in tok])
                continue
            except ValueError:
                pass
        flush()
        head, kinds = tok[0].capitalize(), ("S", "P", "D", "F", "SP")
        if head.upper() in _Z and len(tok) == 2 and tok[1].upper() in kinds:        # NWChem:     "C    S"
            sym, block = head, (tok[1].upper(), [])
        elif head.upper() in _Z and len(tok) == 2 and tok[1] == "0":                # Gaussian94: "C     0"
            sym = head
        elif tok[0].upper() in kinds and sym and len(tok) >= 2:                      # Gaussian94: "S   5   1.00"
            block = (tok[0].upper(), [])
        else:
            raise ValueError(f"{path}: cannot parse line {raw!r}")
    flush()
    if not table:
        raise ValueError(f"{path}: no basis functions found")
    name = (name or os.path.splitext(os.path.basename(path))[0]).lower().replace("_", "-")
    # merged element by element into a basis of that name if there is one (the file's elements win): a file with only
    # the P and S tables on top of the shipped def2-svp keeps H, C, N, O (dft.py --basis def2-svp --basis-file P_S.nw)
    merged = dict(_BASIS_SETS.get(name, {}))
    merged.update(table)
    register_basis(name, merged)
    return name


def _radial_matrices(l, exps, coefs_norm, Z):
    """S, T, V of the one-centre radial problem for primitives r^l exp(-a r^2) (unit-normalised), charge Z."""
    a = np.asarray(exps)[:, None]; b = np.asarray(exps)[None, :]
    n = np.array([gto_norm(l, x) for x in exps])
    nn = n[:, None] * n[None, :]
    p = a + b
    S = nn * np.vectorize(gaussian_int)(2 * l + 2, p)
    # <T> = (2l+3) ab/(a+b) S  for same-l Gaussians;  <1/r> = int r^(2l+1) exp(-p r^2)
    T = (2 * l + 3) * a * b / p * S
    V = -Z * nn * np.vectorize(gaussian_int)(2 * l + 1, p)
    return S, T, V


def check_table(sym, shells, pattern=None):
    """Invariants of one element's table that a typing slip breaks; returns a list of complaints (empty = ok).

    * the contraction pattern is the published one (`pattern`: [(l, nprim), ...]);
    * exponents are positive and strictly decreasing inside a shell, coefficients non-zero;
    * the one-electron atom of the same nuclear charge: the lowest eigenvalue of -1/2 lap - Z/r in the span of
      each l's UNCONTRACTED primitives lies above the exact -Z^2/(2 (l+1)^2) (variational) and, for the occupied
      angular momenta of the neutral atom, within a few percent of it (a misplaced decimal point in a tight
      exponent moves it far outside);
    * the contracted functions of one l are linearly independent (smallest overlap eigenvalue > 1e-6)."""
    from scipy.linalg import eigh as _eigh
    Z = atomic_number(sym)
    bad = []
    if pattern is not None and sorted((l, len(p)) for l, p in shells) != sorted(pattern):
        bad.append(f"{sym}: contraction pattern {sorted((l, len(p)) for l, p in shells)} != published {sorted(pattern)}")
    occupied_l = 0 if Z <= 4 else 1
    for l in sorted({l for l, _ in shells}):
        exps = sorted({e for ll, p in shells if ll == l for e, _ in p}, reverse=True)
        for ll, prims in shells:
            e = [x for x, _ in prims]
            if ll == l and (any(x <= 0 for x in e) or any(e[i] <= e[i + 1] for i in range(len(e) - 1)) or any(c == 0 for _, c in prims)):
                bad.append(f"{sym} l={l}: exponents not positive / strictly decreasing, or a zero coefficient")
        S, T, V = _radial_matrices(l, exps, None, Z)
        e0 = float(_eigh(T + V, S, eigvals_only=True)[0])
        exact = -Z * Z / (2.0 * (l + 1) ** 2)
        if e0 < exact * (1 + 1e-9):
            bad.append(f"{sym} l={l}: hydrogenic ground level {e0:.6f} below the exact {exact:.6f}")
        if l <= occupied_l and e0 > exact * (1 - (0.02 if l == 0 else 0.05)):   # shipped tables: s 0.988-1.000, p 0.969-0.9996 of the exact level
            bad.append(f"{sym} l={l}: hydrogenic ground level {e0:.6f} too far above the exact {exact:.6f}")
        # contracted functions of this l
        cs = [normalized_coefficients(l, [x for x, _ in p], [c for _, c in p]) for ll, p in shells if ll == l]
        es = [[x for x, _ in p] for ll, p in shells if ll == l]
        ov = np.array([[sum(ci * cj * gaussian_int(2 * l + 2, a + b) for a, ci in zip(ea, ca) for b, cj in zip(eb, cb))
                        for eb, cb in zip(es, cs)] for ea, ca in zip(es, cs)])
        if np.linalg.eigvalsh(ov)[0] < 1e-6:
            bad.append(f"{sym} l={l}: contracted functions (nearly) linearly dependent")
    return bad


def gaussian_int(n, alpha):
    """int_0^inf r^n exp(-alpha r^2) dr  (PySCF gto.gaussian_int)."""
    n1 = (n + 1) * 0.5
    return math.gamma(n1) / (2.0 * alpha ** n1)


def gto_norm(l, alpha):
    """Radial normalisation of r^l exp(-alpha r^2)  (PySCF gto.gto_norm)."""
    return 1.0 / math.sqrt(gaussian_int(2 * l + 2, 2.0 * alpha))


def normalized_coefficients(l, exps, coefs):
    """Primitive norm folded in, contraction renormalised to <phi|phi> = 1."""
    exps = np.asarray(exps, dtype=np.float64)
    c = np.asarray(coefs, dtype=np.float64) * np.array([gto_norm(l, a) for a in exps])
    ee = np.array([[gaussian_int(2 * l + 2, a + b) for b in exps] for a in exps])
    return c / math.sqrt(float(c @ ee @ c))


def parse_xyz(text_or_path):
    """XYZ file (2 header lines, Angstrom) or 'El x y z' lines / ';'-separated string.
    Returns ([symbols], coords in bohr (natom,3))."""
    if os.path.exists(text_or_path):
        with open(text_or_path) as f:
            lines = f.readlines()[2:]  # same slice as load_xyz_as_string (dft.py:97-99)
    else:
        lines = text_or_path.replace(";", "\n").splitlines()
    syms, xyz = [], []
    for ln in lines:
        t = ln.split()
        if len(t) < 4:
            continue
        s = t[0].capitalize()
        if s.upper() not in _Z:
            raise ValueError(f"unknown element {t[0]!r}")
        syms.append(s)
        xyz.append([float(t[1]), float(t[2]), float(t[3])])
    if not syms:
        raise ValueError("no atoms found")
    return syms, np.array(xyz) / BOHR


def atomic_number(sym):
    return _Z[sym.upper()]


@dataclass
class ShellTable:
    xyz: np.ndarray      # (nshell, 3) bohr
    l: np.ndarray        # (nshell,)
    nprim: np.ndarray
    off: np.ndarray      # first primitive of each shell
    ao: np.ndarray       # first AO column of each shell
    exp: np.ndarray      # (nprim_total,)
    coef: np.ndarray     # normalised
    atom: np.ndarray     # owning atom of each shell
    nao: int

    @property
    def nshell(self):
        return len(self.l)


def build_shells(symbols, coords_bohr, basis="sto-3g"):
    table = _BASIS_SETS[basis.lower().replace("_", "-")]
    xyz, ls, nprim, off, ao, exps, coefs, owner = [], [], [], [], [], [], [], []
    col = 0
    for ia, (sym, r) in enumerate(zip(symbols, coords_bohr)):
        if sym not in table:
            raise KeyError(f"basis {basis!r} has no entry for element {sym}: register it with basis.load_basis_file(<NWChem or "
                           f"Gaussian94 file>, {basis!r}) (dft.py --basis-file) -- tables not reproducible from memory are not shipped")
        for l, prims in sorted(table[sym], key=lambda t: t[0]):  # PySCF orders shells by l
            e = [p[0] for p in prims]
            c = normalized_coefficients(l, e, [p[1] for p in prims])
            xyz.append(r); ls.append(l); nprim.append(len(e)); off.append(len(exps)); ao.append(col)
            owner.append(ia)
            exps.extend(e); coefs.extend(c.tolist())
            col += 2 * l + 1
    return ShellTable(np.array(xyz, dtype=np.float64), np.array(ls, dtype=np.int32),
                      np.array(nprim, dtype=np.int32), np.array(off, dtype=np.int32),
                      np.array(ao, dtype=np.int32), np.array(exps), np.array(coefs),
                      np.array(owner, dtype=np.int32), col)


def synthetic_shells(symbols, coords_bohr, pattern, seed=20260128):
    """Shell table of a given *shape* with seeded exponents, for benchmarks of
    bases whose tables are not shipped (def2-TZVP, def2-SVP of P/S).
    pattern: {element: [(l, nprim), ...]}; exponents log-uniform in [0.1, 1e3]
    for contracted shells, [0.1, 2] for single primitives."""
    rng = np.random.default_rng(seed)
    table = {}
    for sym in sorted(set(symbols)):
        shells = []
        for l, npr in pattern[sym]:
            hi = 1e3 if npr > 1 else 2.0
            e = np.sort(np.exp(rng.uniform(np.log(0.1), np.log(hi), npr)))[::-1]
            c = rng.uniform(0.2, 1.0, npr)
            shells.append((l, list(zip(e.tolist(), c.tolist()))))
        table[sym] = shells
    name = f"synthetic-{seed}"
    register_basis(name, table)
    return build_shells(symbols, coords_bohr, name)


# contraction patterns (l, nprim) of the Ahlrichs bases used by BASELINE.json's configs
DEF2_SVP_PATTERN = {
    "H": [(0, 3), (0, 1), (1, 1)],
    "C": [(0, 5), (0, 1), (0, 1), (1, 3), (1, 1), (2, 1)],
    "N": [(0, 5), (0, 1), (0, 1), (1, 3), (1, 1), (2, 1)],
    "O": [(0, 5), (0, 1), (0, 1), (1, 3), (1, 1), (2, 1)],
    "P": [(0, 5), (0, 3), (0, 1), (0, 1), (1, 5), (1, 1), (1, 1), (2, 1)],
    "S": [(0, 5), (0, 3), (0, 1), (0, 1), (1, 5), (1, 1), (1, 1), (2, 1)],
}
DEF2_TZVP_PATTERN = {
    "H": [(0, 3), (0, 1), (0, 1), (1, 1)],
    "C": [(0, 6), (0, 2), (0, 1), (0, 1), (0, 1), (1, 4), (1, 1), (1, 1), (2, 1), (2, 1), (3, 1)],
    "N": [(0, 6), (0, 2), (0, 1), (0, 1), (0, 1), (1, 4), (1, 1), (1, 1), (2, 1), (2, 1), (3, 1)],
    "O": [(0, 6), (0, 2), (0, 1), (0, 1), (0, 1), (1, 4), (1, 1), (1, 1), (2, 1), (2, 1), (3, 1)],
}
