"""Shared input recipes for the parity tests."""
import numpy as np

SEED = 20260128


def synth_inputs(ngrid, nao, need_grad=True, seed=SEED, nocc=None):
    """ao = 0.4 N, ao_grad = 0.3 N, w = 0.05 U, C = 0.7 N (nao, nocc), dm = 2 C C^T."""
    rng = np.random.default_rng(seed)
    ao = 0.4 * rng.standard_normal((ngrid, nao))
    gr = 0.3 * rng.standard_normal((3, ngrid, nao)) if need_grad else None
    w = 0.05 * rng.random(ngrid)
    nocc = nocc or max(1, -(-nao // 5))
    C = 0.7 * rng.standard_normal((nao, nocc))
    dm = 2.0 * C @ C.T
    return dm, ao, gr, w
