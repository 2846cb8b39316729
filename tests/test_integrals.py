"""Offline pins of the integral engine (PySCF/libcint itself is not available)."""
import os

import numpy as np
import pytest
from scipy.linalg import eigh

import oracle
from quantum_compute_dft_amd import basis, grid_gen, integrals

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quantum_compute_dft_amd", "data")


def _rhf(S, H, eri, nocc, enuc, iters=60):
    e, C = eigh(H, S)
    dm = 2 * C[:, :nocc] @ C[:, :nocc].T
    E = 0
    for _ in range(iters):
        J = np.einsum("ijkl,kl->ij", eri, dm); K = np.einsum("ikjl,kl->ij", eri, dm)
        F = H + J - 0.5 * K
        E = 0.5 * np.sum(dm * (H + F)) + enuc
        e, C = eigh(F, S)
        dm = 2 * C[:, :nocc] @ C[:, :nocc].T
    return E


def test_h2_sto3g_textbook_values():
    # Szabo & Ostlund, R = 1.4 bohr: S12 0.6593, T11 0.7600, (11|11) 0.7746, (11|22) 0.5697,
    # (21|11) 0.4441, (21|21) 0.2970, E(RHF) -1.1167
    syms, xyz = ["H", "H"], np.array([[0, 0, 0], [0, 0, 1.4]])
    sh = basis.build_shells(syms, xyz, "sto-3g")
    S, T, V = integrals.int1e(sh, syms, xyz)
    eri = integrals.int2e(sh)
    assert S[0, 1] == pytest.approx(0.6593, abs=1e-4) and S[0, 0] == pytest.approx(1.0, abs=1e-12)
    assert T[0, 0] == pytest.approx(0.7600, abs=1e-4) and T[0, 1] == pytest.approx(0.2365, abs=1e-4)
    assert V[0, 0] == pytest.approx(-1.2266 - 0.6538, abs=2e-4)
    assert eri[0, 0, 0, 0] == pytest.approx(0.7746, abs=1e-4)
    assert eri[0, 0, 1, 1] == pytest.approx(0.5697, abs=1e-4)
    assert eri[1, 0, 0, 0] == pytest.approx(0.4441, abs=1e-4)
    assert eri[1, 0, 1, 0] == pytest.approx(0.2970, abs=1e-4)
    E = _rhf(S, T + V, eri, 1, integrals.energy_nuc(syms, xyz))
    assert E == pytest.approx(-1.1167, abs=1e-4)


@pytest.mark.parametrize("bname", ["sto-3g", "def2-svp"])
def test_one_electron_integrals_against_quadrature(bname):
    syms, xyz = basis.parse_xyz(os.path.join(DATA, "H2O.xyz"))
    sh = basis.build_shells(syms, xyz, bname)
    g = grid_gen.Grids(syms, xyz, level=3)
    ao, gr = oracle.eval_ao(sh, g.coords, deriv=1)
    S, T, V = integrals.int1e(sh, syms, xyz)
    w = g.weights
    Sq = ao.T @ (w[:, None] * ao)
    Tq = 0.5 * sum(gr[c].T @ (w[:, None] * gr[c]) for c in range(3))
    assert np.abs(S - Sq).max() < 3e-5 and np.allclose(S, S.T, atol=1e-13)
    assert np.abs(T - Tq).max() < 2e-3 and np.allclose(T, T.T, atol=1e-12)   # core functions: quadrature-limited
    vpot = -sum(basis.atomic_number(s) / np.linalg.norm(g.coords - R, axis=1) for s, R in zip(syms, xyz))
    Vq = ao.T @ ((w * vpot)[:, None] * ao)
    assert np.abs(V - Vq).max() < 5e-3 * max(1.0, np.abs(V).max() / 10) and np.allclose(V, V.T, atol=1e-11)


def test_eri_symmetry_and_tight_gaussian_limit():
    # (ij|kk) with k a very tight normalised s function at R -> sqrt(<k|k>-weighted) point charge:
    # (ij|kk) -> -V_ij of a unit charge at R (k*k integrates to 1 for an s product? use its density norm)
    syms, xyz = basis.parse_xyz("O 0 0 0.1173; H 0 0.7572 -0.4692; H 0 -0.7572 -0.4692")
    tight = 4.0e6
    basis.register_basis("probe", {"O": basis._DEF2_SVP["O"], "H": basis._DEF2_SVP["H"], "He": [(0, [(tight, 1.0)])]})
    R = np.array([[0.3, -0.2, 0.5]])
    syms2, xyz2 = syms + ["He"], np.vstack([xyz, R])
    sh = basis.build_shells(syms2, xyz2, "probe")
    eri = integrals.int2e(sh)
    n = sh.nao
    for perm in ((1, 0, 2, 3), (0, 1, 3, 2), (2, 3, 0, 1), (3, 2, 1, 0)):
        assert np.allclose(eri, eri.transpose(perm), atol=1e-12)
    k = n - 1                                        # the probe function
    S, T, V = integrals.int1e(sh, ["He"], R)         # attraction to a unit... Z=2 charge at R
    skk = S[k, k]
    # density phi_k^2 integrates to 1 (normalised) and is a delta function on this scale
    assert np.allclose(eri[:n - 1, :n - 1, k, k], -V[:n - 1, :n - 1] / 2.0 * skk, rtol=2e-4, atol=2e-6)


def test_water_sto3g_rhf_energy_is_sane():
    # RHF/STO-3G water at the reference's H2O.xyz geometry: literature values for near-equilibrium
    # geometries are -74.96 +- 0.01 Ha
    syms, xyz = basis.parse_xyz(os.path.join(DATA, "H2O.xyz"))
    sh = basis.build_shells(syms, xyz, "sto-3g")
    S, T, V = integrals.int1e(sh, syms, xyz)
    E = _rhf(S, T + V, integrals.int2e(sh), 5, integrals.energy_nuc(syms, xyz))
    assert -74.98 < E < -74.94


def test_pivoted_cholesky_reproduces_the_dense_eri():
    """Integral-direct pivoted Cholesky (cholesky.py): residual below the threshold everywhere,
    symmetric vectors, rank grows with the threshold; column access equals slices of the tensor."""
    from quantum_compute_dft_amd.cholesky import cholesky_eri
    syms, xyz = basis.parse_xyz("O 0 0 0.1173; H 0 0.7572 -0.4692; H 0 -0.7572 -0.4692")
    sh = basis.build_shells(syms, xyz, "def2-svp")
    eri = integrals.int2e(sh)
    n = sh.nao
    cols = integrals.EriColumns(sh)
    assert np.abs(cols.diag() - np.einsum("ijij->ij", eri)).max() < 1e-14
    C, D = 2, sh.nshell - 1                      # an (s|d) style pair, C < D exercises the swapped order
    nc, nd = 2 * sh.l[C] + 1, 2 * sh.l[D] + 1
    ref = eri[:, :, sh.ao[C]:sh.ao[C] + nc, sh.ao[D]:sh.ao[D] + nd].transpose(2, 3, 0, 1).reshape(nc * nd, n, n)
    assert np.abs(cols.cols(C, D, 0.0) - ref).max() < 1e-13
    ranks = []
    for tol in (1e-4, 1e-8):
        L = cholesky_eri(sh, tol=tol)
        ranks.append(L.shape[0])
        assert np.abs(L - L.transpose(0, 2, 1)).max() == 0.0
        assert np.abs(np.einsum("pij,pkl->ijkl", L, L) - eri).max() < tol
    assert ranks[0] < ranks[1] <= n * (n + 1) // 2


def test_def2_tzvp_tables_shape_and_hydrogen_atom():
    """The def2-TZVP tables (H, C; from memory) have the published shapes -- Benzene 222, Anthracene 494
    functions -- and the H table reproduces the published H-atom energy of that basis."""
    from quantum_compute_dft_amd import inputs
    for mol, nao in (("Benzene", 222), ("Anthracene", 494)):
        syms, xyz = basis.parse_xyz(os.path.join(inputs.DATA_DIR, mol + ".xyz"))
        assert basis.build_shells(syms, xyz, "def2-tzvp").nao == nao
    syms, xyz = basis.parse_xyz("H 0 0 0")
    sh = basis.build_shells(syms, xyz, "def2-tzvp")
    S, T, V = integrals.int1e(sh, syms, xyz)
    assert eigh(T + V, S, eigvals_only=True)[0] == pytest.approx(-0.499810, abs=2e-6)


def test_water_rhf_sto3g_literature_anchor():
    """A tighter anchor of the integral engine + STO-3G tables than Szabo-Ostlund's four digits: the water
    molecule of Crawford's "programming projects" (geometry in bohr below; the projects quote
    E_nuc = 8.002367061810450 and E(RHF/STO-3G) = -74.942079928192).  E_nuc confirms the recalled geometry by
    itself; the SCF energy then pins S, T, V, the ERIs and the H / O STO-3G tables to ~1e-8 Ha."""
    from scipy.linalg import eigh
    from quantum_compute_dft_amd import basis, scf
    syms = ["O", "H", "H"]
    xyz = np.array([[0.0, -0.143225816552, 0.0], [1.638036840407, 1.136548822547, 0.0], [-1.638036840407, 1.136548822547, 0.0]])
    assert integrals.energy_nuc(syms, xyz) == pytest.approx(8.002367061810450, abs=1e-10)
    sh = basis.build_shells(syms, xyz, "sto-3g")
    S, T, V = integrals.int1e(sh, syms, xyz)
    eri = integrals.int2e(sh)
    H, nocc = T + V, 5
    e, C = eigh(H, S)
    dm = 2.0 * C[:, :nocc] @ C[:, :nocc].T
    diis, E_old = scf.CDIIS(), 0.0
    for it in range(60):
        J = np.einsum("ijkl,kl->ij", eri, dm)
        K = np.einsum("ikjl,kl->ij", eri, dm)
        F = H + J - 0.5 * K
        E = 0.5 * float(np.sum(dm * (H + F)))
        if abs(E - E_old) < 1e-12:
            break
        E_old = E
        e, C = eigh(diis.update(S, dm, F), S)
        dm = 2.0 * C[:, :nocc] @ C[:, :nocc].T
    assert E + 8.002367061810450 == pytest.approx(-74.942079928192, abs=2e-8)


def test_schwarz_bounds_and_block_pivots_of_the_device_cholesky():
    """Host-side pieces of the factorisation that feeds on DFT_EriColumns: the Schwarz bounds handed to the device kernel
    equal sqrt(max (ab|ab)) over each shell pair of the dense tensor, and the per-block pivoted Cholesky that replaces the
    vector-by-vector loop takes the same pivots and gives the same vectors."""
    from quantum_compute_dft_amd.cholesky import _block_pivots
    syms, xyz = basis.parse_xyz("O 0 0 0.1173; H 0 0.7572 -0.4692; H 0 -0.7572 -0.4692")
    sh = basis.build_shells(syms, np.asarray(xyz), "def2-svp")
    eri = integrals.int2e(sh)
    n = sh.nao
    cols = integrals.EriColumns(sh)
    diag = cols.diag()
    cols.close()
    assert np.allclose(diag, np.einsum("ijij->ij", eri), atol=1e-13)
    q = integrals.schwarz_bounds(sh, diag)
    k = 0
    for a in range(sh.nshell):
        for b in range(a + 1):
            a0, a1 = int(sh.ao[a]), int(sh.ao[a]) + 2 * int(sh.l[a]) + 1
            b0, b1 = int(sh.ao[b]), int(sh.ao[b]) + 2 * int(sh.l[b]) + 1
            assert q[k] == pytest.approx(np.sqrt(max(0.0, diag[a0:a1, b0:b1].max())), rel=1e-14)
            k += 1
    assert k == len(q)
    # Schwarz: |(ab|cd)| <= q_ab q_cd for every quartet of shells
    blk = lambda s: slice(int(sh.ao[s]), int(sh.ao[s]) + 2 * int(sh.l[s]) + 1)
    pid = lambda a, b: a * (a + 1) // 2 + b
    rng = np.random.default_rng(3)
    for _ in range(200):
        a, c = sorted(rng.integers(0, sh.nshell, 2))[::-1], sorted(rng.integers(0, sh.nshell, 2))[::-1]
        v = np.abs(eri[blk(a[0]), blk(a[1]), blk(c[0]), blk(c[1])]).max()
        assert v <= q[pid(*a)] * q[pid(*c)] * (1 + 1e-12) + 1e-15
    # block pivots against the sequential loop on a rank-deficient PSD block with trailing columns
    X = rng.normal(size=(9, 5)); Y = np.vstack([X, rng.normal(size=(6, 5))]); R = X @ Y.T; A = R[:, :9]
    B, G = _block_pivots(A, 1e-10, 100)
    res, d, vs, piv = R.copy(), np.diag(A).copy(), [], []
    while True:
        b = int(np.argmax(d))
        if d[b] < 1e-10:
            break
        v = res[b] / np.sqrt(d[b]); vs.append(v); piv.append(b); d -= v[:9] ** 2; d[b] = 0.0; res -= np.outer(v[:9], v)
    assert B == piv and len(B) == 5 and np.allclose(G, np.tril(G))
    assert np.abs(np.linalg.solve(G, R[B]) - np.array(vs)).max() < 1e-12
    assert _block_pivots(A, 1e-10, 2)[0] == piv[:2]                      # room for two more vectors only


def test_lower_triangular_inverse_of_the_block_factor():
    """cholesky._lower_inverse (the vectors of a step are G^-1 res[B]): exact inverse, lower-triangular, for the sizes a
    joint block of eight shell pairs reaches."""
    from quantum_compute_dft_amd.cholesky import _lower_inverse
    rng = np.random.default_rng(11)
    for r in (1, 2, 9, 49, 200):
        G = np.tril(rng.standard_normal((r, r))) + (2.0 + np.sqrt(r)) * np.eye(r)
        X = _lower_inverse(G)
        assert np.abs(X @ G - np.eye(r)).max() <= 1e-12 and np.abs(np.triu(X, 1)).max() == 0.0
