"""Pins the CPU oracle to the reference: SURVEY.md Appendix D golden values
(outputs of the reference's own kernels:: arithmetic) + internal consistency."""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import synth_inputs


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "appendix_d.json")) as f:
        g = json.load(f)
    g["_inputs"] = np.load(os.path.join(golden_dir, g["whole_path"]["inputs"]))
    return g


def test_pointwise_kats(gold):
    for k in gold["pointwise"]:
        got = oracle.pointwise(k["kind"], [k["rho"]], [k["sigma"]], quirks=True)[0]
        assert got[0] == pytest.approx(k["e"], rel=2e-15, abs=0), k
        if "vrho" in k:
            assert got[1] == pytest.approx(k["vrho"], rel=2e-15, abs=0), k
        if "vsigma" in k:
            assert got[2] == pytest.approx(k["vsigma"], rel=2e-15, abs=0), k
        if "vrho_corrected" in k:
            fix = oracle.pointwise(k["kind"], [k["rho"]], [k["sigma"]], quirks=False)[0]
            assert fix[1] == pytest.approx(k["vrho_corrected"], rel=2e-10), k
            assert fix[0] == got[0]  # energies untouched by the derivative fix


@pytest.mark.parametrize("name", ["LDA", "GGA", "B3LYP"])
def test_whole_path_golden(gold, name):
    wp = gold["whole_path"]; ref = wp[name]; d = gold["_inputs"]
    exc, v, rho, _ = oracle.compute_xc(ref["type"], d["dm"], d["ao"], d["weights"],
                                        d["ao_grad"], quirks=True, want_density=True)
    assert float(d["weights"] @ rho) == pytest.approx(wp["sum_w_rho"], rel=1e-12)
    assert exc == pytest.approx(ref["exc"], rel=1e-15, abs=0)
    assert np.linalg.norm(v) == pytest.approx(ref["vxc_fro"], rel=1e-15, abs=0)
    if "v01" in ref:
        assert v[0, 1] == pytest.approx(ref["v01"], rel=1e-15)
        assert v[1, 0] == pytest.approx(ref["v10"], rel=1e-15)
    if ref.get("symmetric"):
        assert np.array_equal(v, v.T)


def _fd(kind, rho, sigma, quirks):
    """d(rho*e)/drho and d(rho*e)/dsigma by central differences."""
    h = 1e-6 * rho
    f = lambda r, s: r * oracle.pointwise(kind, [r], [s], quirks=quirks)[0, 0]
    dr = (f(rho + h, sigma) - f(rho - h, sigma)) / (2 * h)
    hs = 1e-6 * max(sigma, 1e-8)
    ds = (f(rho, sigma + hs) - f(rho, sigma - hs)) / (2 * hs)
    return dr, ds


@pytest.mark.parametrize("kind", ["slater", "vwn5", "vwn_rpa", "pw92", "pbe_x", "pbe_c", "b88", "lyp"])
def test_corrected_derivatives_match_finite_differences(kind):
    for rho, sigma in [(0.1, 0.05), (1.0, 2.0), (1e-3, 1e-5), (5.0, 1.0)]:
        out = oracle.pointwise(kind, [rho], [sigma], quirks=False)[0]
        dr, ds = _fd(kind, rho, sigma, False)
        assert out[1] == pytest.approx(dr, rel=2e-6, abs=1e-9), (kind, rho, sigma)
        if kind in ("pbe_x", "pbe_c", "b88", "lyp"):
            assert out[2] == pytest.approx(ds, rel=2e-6, abs=1e-9), (kind, rho, sigma)


def test_quirks_only_touch_the_two_documented_derivatives():
    rho = np.array([1e-3, 0.1, 1.0, 50.0]); sig = np.array([1e-3, 0.05, 2.0, 9.0])
    for kind in ["slater", "vwn_rpa", "pw92", "pbe_x", "b88", "lyp", "b3lyp"]:
        assert np.array_equal(oracle.pointwise(kind, rho, sig, True), oracle.pointwise(kind, rho, sig, False))
    for kind in ["vwn5", "pbe_c", "lda", "gga"]:
        a, b = oracle.pointwise(kind, rho, sig, True), oracle.pointwise(kind, rho, sig, False)
        assert np.array_equal(a[:, 0], b[:, 0]) and np.array_equal(a[:, 2], b[:, 2])
        assert not np.allclose(a[:, 1], b[:, 1], rtol=1e-6, atol=0)


def test_density_cutoffs():
    for kind in oracle.POINTWISE_KINDS:
        # LYP alone cuts at 1e-14 (dft_solver.cu:144); unreachable behind the kernel guard :447
        lo = 9e-15 if kind == "lyp" else 9e-13
        assert np.all(oracle.pointwise(kind, [0.0, lo, -1.0], [1.0, 1.0, 1.0]) == 0.0)
    assert np.any(oracle.pointwise("lyp", [9e-13], [1e-30]) != 0.0)
    assert np.all(oracle.pointwise("b3lyp", [9e-13], [1.0]) == 0.0)
    # B88 zero-gradient guard (dft_solver.cu:80)
    assert np.all(oracle.pointwise("b88", [0.3], [1e-21]) == 0.0)


@pytest.mark.parametrize("xc_type", [0, 1, 2])
def test_sweep_against_numpy_einsum(xc_type):
    dm, ao, gr, w = synth_inputs(257, 13)
    exc, v, rho, grad = oracle.compute_xc(xc_type, dm, ao, w, gr, want_density=True)
    x = ao @ dm
    assert np.allclose(rho, np.einsum("gi,gi->g", x, ao), rtol=1e-12, atol=1e-14)
    if xc_type == 0:
        pw = oracle.pointwise("lda", rho)
        B = (w * pw[:, 1])[:, None] * ao
        V = B.T @ ao
    else:
        g3 = 2.0 * np.einsum("gi,cgi->gc", x, gr)
        assert np.allclose(grad, g3, rtol=1e-11, atol=1e-13)
        sigma = np.einsum("gc,gc->g", grad, grad)
        pw = oracle.pointwise("gga" if xc_type == 1 else "b3lyp", rho, sigma)
        k = 4.0 if xc_type == 1 else 2.0
        dot = np.einsum("gc,cgi->gi", grad, gr)
        B = w[:, None] * (pw[:, 1:2] * ao + k * pw[:, 2:3] * dot)
        V = B.T @ ao
        if xc_type == 2:
            V = V + V.T
    assert exc == pytest.approx(float(w @ pw[:, 0]), rel=1e-13)
    assert np.allclose(v, V, rtol=1e-11, atol=1e-13)


def test_coulomb_and_exchange_match_einsum():
    rng = np.random.default_rng(5)
    n = 6
    eri = rng.standard_normal((n, n, n, n))
    dm = rng.standard_normal((n, n))
    J = oracle.coulomb(eri, dm)
    # cublasDgemv(OP_N) on the row-major buffer seen column-major = ERI^T . vec(D)
    assert np.allclose(J, np.einsum("klij,kl->ij", eri, dm), rtol=1e-13, atol=1e-13)
    K = oracle.exchange(eri, dm)
    assert np.allclose(K, np.einsum("ijkl,jl->ik", eri, dm), rtol=1e-13, atol=1e-13)
