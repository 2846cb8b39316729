"""Becke/Lebedev molecular integration grids -- the recipe the reference gets from PySCF at
grid.py:33-36 (`dft.gen_grid.Grids(mol); grids.level = 3; grids.build()`), SURVEY.md App. B:

* radial: Treutler-Ahlrichs M4 map, xi = 1,  r = -(1/ln2) (1+x)^0.6 ln((1-x)/2) on Chebyshev-2
  nodes, n_rad = 50 / 75 / 80 for periods 1 / 2 / 3 at level 3, weight 4 pi r^2 dr;
* angular: Lebedev 302 (periods 1-2) / 434 (period 3) points, NWChem pruning by Bragg radius
  (region edges 0.25,0.5,1.0,4.5 | 0.1667,0.5,0.9,3.5 | 0.1,0.4,0.8,2.5 -> 50, 86, n-1, n, n-1);
* Becke partition, three smoothing iterations, Treutler radius adjustment
  a_ij = (sqrt(R_i/R_j) - sqrt(R_j/R_i))/4 clipped to +-1/2.

PARITY UNPINNED against PySCF itself (not installed; recipe from memory); pinned offline by the
known point counts (H2O 34 310, Benzene 143 556, Anthracene 294 868) and by
exact integrals of atom-centred Gaussians (tests/test_grid_gen.py).  Lebedev rules come from
scipy.integrate.lebedev_rule.  The Becke weights are evaluated with torch so the O(ngrid*natom^2)
part can run on the GPU for the large BASELINE molecules.
"""
import math

import numpy as np
import torch

from .basis import BOHR, atomic_number

# Bragg-Slater radii (Angstrom) as PySCF's radi.BRAGG_RADII, elements used by the BASELINE configs
_BRAGG = {1: 0.35, 2: 1.40, 3: 1.45, 4: 1.05, 5: 0.85, 6: 0.70, 7: 0.65, 8: 0.60, 9: 0.50, 10: 1.50,
          11: 1.80, 12: 1.50, 13: 1.25, 14: 1.10, 15: 1.00, 16: 1.00, 17: 1.00, 18: 1.80}
_LEBEDEV_NGRID = [1, 6, 14, 26, 38, 50, 74, 86, 110, 146, 170, 194, 230, 266, 302, 350, 434, 590]
_LEBEDEV_DEGREE = {6: 3, 14: 5, 26: 7, 38: 9, 50: 11, 74: 13, 86: 15, 110: 17, 146: 19, 170: 21,
                   194: 23, 230: 25, 266: 27, 302: 29, 350: 31, 434: 35, 590: 41}
# level -> (radial points, Lebedev points) per period 1, 2, 3 (PySCF RAD_GRIDS / ANG_ORDER)
_LEVELS = {
    0: ((10, 15, 20), (50, 86, 110)),
    1: ((30, 40, 50), (110, 146, 170)),
    2: ((40, 60, 65), (194, 266, 266)),
    3: ((50, 75, 80), (302, 302, 434)),
    4: ((60, 90, 95), (434, 590, 590)),
}
_LEB_CACHE = {}


def _period(z):
    return 1 if z <= 2 else 2 if z <= 10 else 3


def lebedev(npts):
    if npts not in _LEB_CACHE:
        from scipy.integrate import lebedev_rule
        x, w = lebedev_rule(_LEBEDEV_DEGREE[npts])
        assert x.shape[1] == npts
        _LEB_CACHE[npts] = (x.T.copy(), w.copy())  # weights sum to 4 pi
    return _LEB_CACHE[npts]


def treutler_ahlrichs(n):
    """r_i, dr_i ascending in r (PySCF radi.treutler_ahlrichs)."""
    i = np.arange(1, n + 1)
    step = math.pi / (n + 1)
    x = np.cos(i * step)
    ln2 = 1.0 / math.log(2.0)
    r = -ln2 * (1 + x) ** 0.6 * np.log((1 - x) / 2)
    dr = step * np.sin(i * step) * ln2 * (1 + x) ** 0.6 * (-0.6 / (1 + x) * np.log((1 - x) / 2) + 1 / (1 - x))
    return r[::-1].copy(), dr[::-1].copy()


def nwchem_prune(z, rads, n_ang):
    alphas = np.array(((0.25, 0.5, 1.0, 4.5), (0.1667, 0.5, 0.9, 3.5), (0.1, 0.4, 0.8, 2.5)))
    leb = np.array(_LEBEDEV_NGRID[4:])
    if n_ang < 50:
        return np.repeat(n_ang, len(rads))
    if n_ang == 50:
        leb_l = np.array([1, 2, 2, 2, 1])
    else:
        idx = int(np.where(leb == n_ang)[0][0])
        leb_l = np.array([1, 3, idx - 1, idx, idx - 1])
    r_atom = _BRAGG[z] / BOHR + 1e-200
    row = 0 if z <= 2 else 1 if z <= 10 else 2
    place = ((rads / r_atom).reshape(-1, 1) > alphas[row]).sum(axis=1)
    return leb[leb_l[place]]


def atomic_grid(z, level=3):
    """Points (relative to the nucleus, bohr) and weights of one atom's grid."""
    nrad, nang = _LEVELS[level][0][_period(z) - 1], _LEVELS[level][1][_period(z) - 1]
    r, dr = treutler_ahlrichs(nrad)
    wr = 4.0 * math.pi * r * r * dr
    angs = nwchem_prune(z, r, nang)
    pts, wts = [], []
    for ri, wi, n in zip(r, wr, angs):
        x, w = lebedev(int(n))
        pts.append(ri * x)
        wts.append(wi * w / (4.0 * math.pi))
    return np.concatenate(pts), np.concatenate(wts)


def becke_weights(coords, atom_index, atom_xyz, charges, device="cpu", block=65536):
    """Becke cell function of each point's own atom, normalised (PySCF original_becke +
    treutler_atomic_radii_adjust).  coords (n,3), atom_index (n,), atom_xyz (natm,3)."""
    dev = torch.device(device)
    R = torch.as_tensor(atom_xyz, dtype=torch.float64, device=dev)
    natm = R.shape[0]
    rad = torch.sqrt(torch.tensor([_BRAGG[int(z)] / BOHR for z in charges], dtype=torch.float64, device=dev)) + 1e-200
    rr = rad[:, None] / rad[None, :]
    a = (0.25 * (rr.T - rr)).clamp(-0.5, 0.5)
    Rij = torch.cdist(R, R)
    Rij.fill_diagonal_(1.0)
    out = np.empty(len(coords))
    C = torch.as_tensor(coords, dtype=torch.float64, device=dev)
    own = torch.as_tensor(atom_index, dtype=torch.long, device=dev)
    for lo in range(0, len(coords), block):
        c = C[lo:lo + block]
        d = torch.cdist(c, R)                                    # (nb, natm)
        mu = (d[:, :, None] - d[:, None, :]) / Rij[None]         # (nb, i, j)
        g = mu + a[None] * (1 - mu * mu)
        for _ in range(3):
            g = (3 - g * g) * g * 0.5
        s = 0.5 * (1 - g)
        eye = torch.eye(natm, dtype=torch.bool, device=dev)
        s = torch.where(eye[None], torch.ones_like(s), s)
        p = s.prod(dim=2)                                         # cell function of every atom
        w = p.gather(1, own[lo:lo + block, None])[:, 0] / p.sum(dim=1)
        out[lo:lo + block] = w.cpu().numpy()
    return out


class Grids:
    """`coords` (ngrid,3) bohr and `weights` (ngrid,), like PySCF's Grids after build()."""

    def __init__(self, symbols, coords_bohr, level=3, device="cpu"):
        self.symbols, self.atom_xyz, self.level = list(symbols), np.asarray(coords_bohr, dtype=np.float64), level
        charges = [atomic_number(s) for s in symbols]
        pts, wts, own = [], [], []
        cache = {}
        for ia, z in enumerate(charges):
            if z not in cache:
                cache[z] = atomic_grid(z, level)
            p, w = cache[z]
            pts.append(p + self.atom_xyz[ia])
            wts.append(w)
            own.append(np.full(len(w), ia))
        self.coords = np.concatenate(pts)
        own = np.concatenate(own)
        w0 = np.concatenate(wts)
        if len(charges) > 1:
            w0 = w0 * becke_weights(self.coords, own, self.atom_xyz, charges, device=device)
        self.weights = w0
        self.atom_index = own

    @property
    def size(self):
        return len(self.weights)
