"""DFT_ComputeXCOcc -- the density step through the occupied orbitals (csrc/xc_occ_kernels.hpp) -- against the CPU
oracle fed with dm = cocc cocc^T (the reference contracts the full matrix: dft_solver.cu:294-307, 346-380; the driver
holds the orbitals, dft.py:181-182).  Same tolerances as test_gpu_parity.py: Exc rel 1e-12, Vxc 1e-11 max|V|."""
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


def occ_inputs(ngrid, nao, nocc, seed):
    """SURVEY 8(d) recipe with the orbitals kept: ao = 0.4 N, grad = 0.3 N, w = 0.05 U, cocc = sqrt(2) 0.7 N."""
    rng = np.random.default_rng(seed)
    ao = 0.4 * rng.standard_normal((ngrid, nao))
    gr = 0.3 * rng.standard_normal((3, ngrid, nao))
    w = 0.05 * rng.random(ngrid)
    cocc = np.sqrt(2.0) * 0.7 * rng.standard_normal((nao, nocc))
    return cocc, cocc @ cocc.T, ao, gr, w


def _solver(xc_type, **opts):
    s = q.DFTSolverWrapper(q.build_library(), NAMES[xc_type])
    for k, v in opts.items():
        s.set_option(k, v)
    return s


def _run(s, xc_type, cocc, dm, ao, gr, w, dev, pass_dm=True):
    ngrid, nao = ao.shape
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    d_c, d_dm, d_ao, d_w = t(cocc), t(dm), t(ao), t(w)
    d_gr = t(gr) if xc_type else None
    d_v = torch.full((nao, nao), 7.0, dtype=torch.float64, device=dev)
    exc = s.compute_xc_occ(ngrid, nao, cocc.shape[1], d_c, d_ao, d_w, d_v, d_gr, d_dm if pass_dm else None)
    torch.cuda.synchronize()
    return exc, d_v.cpu().numpy()


def _check(exc, v, exc_ref, v_ref):
    assert exc == pytest.approx(exc_ref, rel=1e-12, abs=1e-14)
    assert np.abs(v - v_ref).max() <= 1e-11 * np.abs(v_ref).max() + 1e-13


# (ngrid, nao, nocc): every orbital-tile count of the resident kernels (nto 1..4), ragged grids and AO widths around
# the 32-column chunk edges, the streamed kernels (C does not fit in LDS: nao 200+), the eight-wave streamed
# kernels (nto 5..8), two passes over the planes (nocc > 128), odd nao (8-byte loads)
OCC_SHAPES = [(1, 1, 1), (7, 3, 2), (96, 5, 2), (257, 13, 3), (1000, 16, 5), (1025, 17, 16), (4097, 24, 5),
              (3001, 36, 18), (2000, 64, 33), (1531, 65, 13), (2500, 114, 21), (1300, 128, 64), (999, 96, 49),
              (700, 129, 26), (900, 200, 40), (1111, 246, 47), (300, 257, 65), (2100, 301, 97), (130, 494, 47),
              (650, 494, 128), (400, 320, 129), (200, 610, 250)]


@pytest.mark.parametrize("ngrid,nao,nocc", OCC_SHAPES)
@pytest.mark.parametrize("xc_type", [0, 1, 2])
def test_occupied_orbital_sweep_matches_oracle(dev, xc_type, ngrid, nao, nocc):
    cocc, dm, ao, gr, w = occ_inputs(ngrid, nao, nocc, seed=4000 + ngrid + nao + nocc)
    exc_ref, v_ref = oracle.compute_xc(xc_type, dm, ao, w, gr if xc_type else None)
    s = _solver(xc_type, occ=1)                     # the occupied form whatever its MFMA count
    exc, v = _run(s, xc_type, cocc, dm, ao, gr, w, dev)
    _check(exc, v, exc_ref, v_ref)


@pytest.mark.parametrize("xc_type", [0, 1, 2])
def test_auto_mode_and_missing_dm(dev, xc_type):
    """Default option: the library picks the path by MFMA count; without a dm it forms cocc cocc^T itself where the
    dm kernels are taken (minimal-basis ratio nocc/nao ~ 0.6), and option occ = 2 never uses the orbitals."""
    for ngrid, nao, nocc in ((1500, 114, 21), (1500, 36, 21), (800, 246, 47), (800, 160, 90)):
        cocc, dm, ao, gr, w = occ_inputs(ngrid, nao, nocc, seed=77 + nao)
        exc_ref, v_ref = oracle.compute_xc(xc_type, dm, ao, w, gr if xc_type else None)
        for opts, pass_dm in (({}, True), ({}, False), ({"occ": 2}, True), ({"occ": 2}, False)):
            exc, v = _run(_solver(xc_type, **opts), xc_type, cocc, dm, ao, gr, w, dev, pass_dm=pass_dm)
            _check(exc, v, exc_ref, v_ref)


def test_occupied_form_equals_dm_form_at_benzene_size(dev):
    """Full Benzene/def2-SVP shape (143 556 x 114, 21 occupied): the occupied-orbital call against the dm call of the
    same library (the oracle takes minutes at this size), plus a slice against the oracle."""
    ngrid, nao, nocc = 143556, 114, 21
    g = torch.Generator(device=dev); g.manual_seed(5)
    ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
    c = 0.7 * np.sqrt(2.0) * torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
    dm = (c @ c.T).contiguous()
    for xc_type in (1, 2):
        s = _solver(xc_type)
        v0 = torch.zeros((nao, nao), dtype=torch.float64, device=dev); v1 = torch.zeros_like(v0)
        e0 = s.compute_xc(ngrid, nao, dm, ao, w, v0, gr)
        e1 = s.compute_xc_occ(ngrid, nao, nocc, c, ao, w, v1, gr, dm)
        torch.cuda.synchronize()
        assert e1 == pytest.approx(e0, rel=1e-12)
        assert float((v1 - v0).abs().max()) <= 1e-11 * float(v0.abs().max())
        n = 4000
        exc_ref, v_ref = oracle.compute_xc(xc_type, dm.cpu().numpy(), ao[-n:].cpu().numpy(), w[-n:].cpu().numpy(),
                                           np.ascontiguousarray(gr[:, -n:].cpu().numpy()), omp=True)
        v2 = torch.zeros_like(v0)
        e2 = s.compute_xc_occ(n, nao, nocc, c, ao[-n:].contiguous(), w[-n:].contiguous(), v2, gr[:, -n:].contiguous())
        _check(e2, v2.cpu().numpy(), exc_ref, v_ref)


def test_async_form_and_determinism(dev):
    cocc, dm, ao, gr, w = occ_inputs(5000, 114, 21, seed=9)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    d_c, d_ao, d_gr, d_w = t(cocc), t(ao), t(gr), t(w)
    s = _solver(1)
    outs = []
    for _ in range(3):
        d_v = torch.zeros((114, 114), dtype=torch.float64, device=dev)
        d_e = torch.zeros(1, dtype=torch.float64, device=dev)
        assert s.compute_xc_occ_async(5000, 114, 21, d_c, d_ao, d_w, d_v, d_e, d_gr) == 0
        torch.cuda.synchronize()
        outs.append((float(d_e.item()), d_v.clone()))
    exc_ref, v_ref = oracle.compute_xc(1, dm, ao, w, gr)
    _check(outs[0][0], outs[0][1].cpu().numpy(), exc_ref, v_ref)
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][1], outs[2][1])


def test_occ_call_without_orbitals_is_an_error(dev):
    s = _solver(1)
    d = torch.zeros((16, 16), dtype=torch.float64, device=dev)
    with pytest.raises(RuntimeError):
        s.compute_xc_occ(16, 16, 4, None, d, d[0], d, torch.zeros((3, 16, 16), dtype=torch.float64, device=dev))


def test_recorded_graph_replay_is_bit_identical_and_follows_new_inputs():
    """Option "graph": the third and later calls with the same pointers are one hipGraphLaunch; results are those of the
    plain launches bit for bit, new values behind the same pointers are picked up, a new pointer takes plain launches again."""
    import torch
    import quantum_compute_dft_amd as q
    dev = torch.device("cuda:0")
    ngrid, nao, nocc = 5000, 24, 5
    g = torch.Generator(device=dev); g.manual_seed(7)
    ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g)
    w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
    c = torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
    dm = (c @ c.T).contiguous()
    for xc in ("LDA", "GGA", "B3LYP"):
        s = q.DFTSolverWrapper(q.library_path(), xc)
        grad = None if xc == "LDA" else gr
        ref = {}
        s.set_option("graph", 0)
        for scale in (1.0, 0.5):
            d = (scale * dm).contiguous(); v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
            ref[scale] = (s.compute_xc(ngrid, nao, d, ao, w, v, grad), v.clone())
        for mode in (1, -1):
            s.set_option("graph", mode)
            d = dm.clone(); v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
            for rep in range(4):                        # plain, recorded, replayed, replayed
                v.zero_()
                e = s.compute_xc(ngrid, nao, d, ao, w, v, grad)
                assert e == ref[1.0][0] and torch.equal(v, ref[1.0][1]), (xc, mode, rep)
            d.mul_(0.5)                                 # same pointers, new density: the replay reads it
            e = s.compute_xc(ngrid, nao, d, ao, w, v, grad)
            assert e == ref[0.5][0] and torch.equal(v, ref[0.5][1]), (xc, mode)
            v2 = torch.zeros_like(v)                    # another output pointer: a new key
            e = s.compute_xc(ngrid, nao, d, ao, w, v2, grad)
            assert e == ref[0.5][0] and torch.equal(v2, ref[0.5][1]), (xc, mode)
            # a larger call in between grows the workspace: the recorded graphs are dropped, not replayed stale
            big = torch.cat([ao, ao, ao]); bw = torch.cat([w, w, w]); bg = None if grad is None else torch.cat([gr, gr, gr], dim=1).contiguous()
            vb = torch.zeros_like(v)
            eb = s.compute_xc(3 * ngrid, nao, d, big, bw, vb, bg)
            assert abs(eb - 3 * ref[0.5][0]) <= 1e-11 * abs(eb)
            for rep in range(3):
                v.zero_()
                e = s.compute_xc(ngrid, nao, d, ao, w, v, grad)
                assert e == ref[0.5][0] and torch.equal(v, ref[0.5][1]), (xc, mode, "after growth", rep)
        cs = (np.sqrt(0.5) * c).contiguous()            # the occupied entry through the same mechanism
        s.set_option("graph", 1)
        for rep in range(3):
            v.zero_()
            e = s.compute_xc_occ(ngrid, nao, nocc, cs, ao, w, v, grad, d)
            assert abs(e - ref[0.5][0]) <= 1e-12 * abs(e) and (v - ref[0.5][1]).abs().max() <= 1e-11 * ref[0.5][1].abs().max()
