"""The one-pass sweep kernel for small bases (nao <= 32, csrc/xc_tiny_kernels.hpp) through DFT_ComputeXC /
DFT_ComputeXCOcc, against the CPU oracle (src/dft_solver.cu:559-672 restated) and against the four-launch path of
the same library (option tiny = 0).  Tolerances as in test_gpu_parity.py: Exc rel 1e-12, Vxc 1e-11 max|V|."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)
import quantum_compute_dft_amd as q  # noqa: E402

NAMES = {0: "LDA", 1: "GGA", 2: "B3LYP"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def inputs(ngrid, nao, seed, symmetric=True):
    """SURVEY 8(d) recipe: ao = 0.4 N, grad = 0.3 N, w = 0.05 U, dm = 0.7^2 * 2 C C^T-like."""
    rng = np.random.default_rng(seed)
    ao = 0.4 * rng.standard_normal((ngrid, nao))
    gr = 0.3 * rng.standard_normal((3, ngrid, nao))
    w = 0.05 * rng.random(ngrid)
    nocc = max(1, nao // 3)
    c = np.sqrt(2.0) * 0.7 * rng.standard_normal((nao, nocc))
    dm = c @ c.T
    if not symmetric:
        dm = dm + 0.05 * rng.standard_normal((nao, nao))
    return c, dm, ao, gr, w


def _solver(xc_type, **opts):
    s = q.DFTSolverWrapper(q.build_library(), NAMES[xc_type])
    for k, v in opts.items():
        s.set_option(k, v)
    return s


def _run(s, xc_type, dm, ao, gr, w, dev):
    ngrid, nao = ao.shape
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    d_dm, d_ao, d_w = t(dm), t(ao), t(w)
    d_gr = t(gr) if xc_type else None
    d_v = torch.full((nao, nao), 7.0, dtype=torch.float64, device=dev)
    exc = s.compute_xc(ngrid, nao, d_dm, d_ao, d_w, d_v, d_gr)
    torch.cuda.synchronize()
    return exc, d_v.cpu().numpy()


def _check(exc, v, exc_ref, v_ref):
    assert exc == pytest.approx(exc_ref, rel=1e-12, abs=1e-14)
    assert np.abs(v - v_ref).max() <= 1e-11 * np.abs(v_ref).max() + 1e-13


def _timing_names(s):
    return [n for n, _ in s.timings()]


# every AO width class (one and two column tiles, odd and even, the tile edges 15/16/17 and 31/32), grids below one
# sub-tile, ragged, more sub-tiles than waves in flight (one workgroup per CU x 8 waves x 16 points = 32 768)
SHAPES = [(1, 1), (1, 5), (7, 3), (15, 2), (16, 16), (17, 15), (33, 17), (96, 5), (257, 13), (1000, 16), (1025, 17),
          (4097, 24), (3001, 31), (2000, 32), (34310, 24), (70001, 7), (150017, 19)]


@pytest.mark.parametrize("ngrid,nao", SHAPES)
@pytest.mark.parametrize("xc_type", [0, 1, 2])
def test_one_pass_sweep_matches_oracle(dev, xc_type, ngrid, nao):
    _, dm, ao, gr, w = inputs(ngrid, nao, seed=9000 + ngrid + nao)
    exc_ref, v_ref = oracle.compute_xc(xc_type, dm, ao, w, gr if xc_type else None, omp=ngrid > 20000)
    s = _solver(xc_type, profile=1, tiny=1)          # on at every size (auto leaves a band of grid sizes to the four launches)
    exc, v = _run(s, xc_type, dm, ao, gr, w, dev)
    assert "sweep_tiny" in _timing_names(s)          # the kernel under test is the one that ran
    _check(exc, v, exc_ref, v_ref)
    if xc_type == 2:
        assert np.array_equal(v, v.T)                # symmetrize_matrix_kernel: bitwise


@pytest.mark.parametrize("xc_type", [0, 1, 2])
def test_one_pass_sweep_against_the_four_launch_path(dev, xc_type):
    """Same library, option tiny = 0: rho -> xc_points -> vxc -> reduce.  Different summation orders, same numbers to
    the conditioning of the sums; quirks on and off; a density matrix that is not symmetric (the reference
    contracts (D + D^T)/2 implicitly through the symmetric product, dft_solver.cu:294-307)."""
    for ngrid, nao, sym in ((5000, 24, True), (5000, 24, False), (777, 9, False), (12345, 32, True)):
        _, dm, ao, gr, w = inputs(ngrid, nao, seed=31 + nao, symmetric=sym)
        for quirks in (1, 0):
            s1 = _solver(xc_type, quirks=quirks, profile=1, tiny=1)
            s0 = _solver(xc_type, quirks=quirks, tiny=0, profile=1)
            e1, v1 = _run(s1, xc_type, dm, ao, gr, w, dev)
            e0, v0 = _run(s0, xc_type, dm, ao, gr, w, dev)
            assert "sweep_tiny" in _timing_names(s1) and "sweep_tiny" not in _timing_names(s0)
            assert e1 == pytest.approx(e0, rel=1e-12)
            assert np.abs(v1 - v0).max() <= 1e-11 * np.abs(v0).max()
            exc_ref, v_ref = oracle.compute_xc(xc_type, dm, ao, w, gr if xc_type else None, quirks=bool(quirks))
            _check(e1, v1, exc_ref, v_ref)


def test_cutoffs_and_negative_density(dev):
    """Points below the density cut-off, exactly zero rows, negative rho (an indefinite D): the per-point bodies'
    branches (dft_solver.cu:12-13 thresholds) inside the one-pass kernel."""
    rng = np.random.default_rng(5)
    ngrid, nao = 3000, 20
    _, dm, ao, gr, w = inputs(ngrid, nao, seed=77)
    ao[::7] *= 1e-8          # rho ~ 1e-16: below kRhoCut
    ao[5::11] = 0.0          # rho = 0 exactly
    gr[:, 3::13] *= 1e-12    # sigma below kSigmaCut
    dm_neg = dm - 3.0 * np.eye(nao)
    for xc_type in (0, 1, 2):
        for d in (dm, dm_neg):
            exc_ref, v_ref = oracle.compute_xc(xc_type, d, ao, w, gr if xc_type else None)
            exc, v = _run(_solver(xc_type, tiny=1), xc_type, d, ao, gr, w, dev)
            assert np.isfinite(exc) and np.isfinite(v).all()
            _check(exc, v, exc_ref, v_ref)


def test_bitwise_reproducible_and_graph_replay(dev):
    _, dm, ao, gr, w = inputs(34310, 24, seed=3)
    for xc_type in (0, 1, 2):
        outs = []
        for opts in ({"graph": 0}, {"graph": 0}, {"graph": 1}):
            s = _solver(xc_type, tiny=1, **opts)
            for _ in range(3):                       # the third call of a graph solver is a replay
                e, v = _run(s, xc_type, dm, ao, gr, w, dev)
            outs.append((e, v))
        for e, v in outs[1:]:
            assert e == outs[0][0] and np.array_equal(v, outs[0][1])


def test_auto_rule(dev):
    """Default option (xc_tiny.hip::tiny_pays): the one-pass kernel wherever it applies (nao <= 32), also right above one
    sub-tile per wave of one workgroup per CU (R) where a partial second round is dealt one sub-tile per CU."""
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    R = ncu * 8 * 16
    for xc_type, nao, ngrid, want in ((1, 16, R + 5000, True), (1, 7, 1000, True), (1, 24, R - 7, True), (1, 24, R + 16, True),
                                      (2, 32, int(1.25 * R), True), (0, 24, int(1.2 * R), True), (1, 32, 4 * R, True),
                                      (1, 33, 1000, False), (0, 40, R, False)):
        _, dm, ao, gr, w = inputs(ngrid, nao, seed=1)
        exc_ref, v_ref = oracle.compute_xc(xc_type, dm, ao, w, gr if xc_type else None, omp=True)
        s = _solver(xc_type, profile=1)
        exc, v = _run(s, xc_type, dm, ao, gr, w, dev)
        assert ("sweep_tiny" in _timing_names(s)) == want, (xc_type, nao, ngrid)
        _check(exc, v, exc_ref, v_ref)      # partial second and later rounds of sub-tiles against the oracle as well


def test_occupied_entry_takes_the_one_pass_kernel(dev):
    """DFT_ComputeXCOcc at nao <= 32: with or without a dm (formed from the orbitals on the device)."""
    ngrid, nao = 6000, 24
    c, dm, ao, gr, w = inputs(ngrid, nao, seed=12)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    for xc_type in (0, 1, 2):
        exc_ref, v_ref = oracle.compute_xc(xc_type, dm, ao, w, gr if xc_type else None)
        for pass_dm in (True, False):
            s = _solver(xc_type, profile=1)
            d_v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
            exc = s.compute_xc_occ(ngrid, nao, c.shape[1], t(c), t(ao), t(w), d_v, t(gr) if xc_type else None,
                                   t(dm) if pass_dm else None)
            torch.cuda.synchronize()
            assert "sweep_tiny" in _timing_names(s)
            _check(exc, d_v.cpu().numpy(), exc_ref, v_ref)


def test_scf_energy_with_and_without_the_one_pass_kernel(dev):
    """H2O/def2-SVP (24 functions, 34 310 points: every sweep of the SCF goes through k_sweep_tiny by default), converged far
    below the driver's thresholds: same energies as with the four-launch sweep."""
    from quantum_compute_dft_amd import inputs as inp_mod, scf
    inp = inp_mod.build("H2O", "def2-svp", 3, verbose=False)
    for fn in ("LDA", "GGA"):
        res = []
        for tiny in (1, 0):
            be = scf.HipBackend(inp, fn)
            be.solver.set_option("tiny", tiny)
            res.append(scf.run_scf(inp, be, fn, log=None, conv_e=1e-11, conv_dm=1e-9))
        assert res[0]["converged"] and res[1]["converged"]
        assert res[0]["E_tot"] == pytest.approx(res[1]["E_tot"], abs=1e-9)
        assert res[0]["E_xc"] == pytest.approx(res[1]["E_xc"], abs=1e-9)
        assert np.abs(res[0]["dm"] - res[1]["dm"]).max() < 1e-7
