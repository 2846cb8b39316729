"""N>1 path on CPU: world_size-2 gloo run of the grid partition + all-reduce of [Vxc|Exc].
The local sweep is the CPU oracle here (tests may use it); on a GPU box it is the HIP solver."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from helpers import synth_inputs
from quantum_compute_dft_amd.grid_shard import ShardedXC, shard_bounds


def test_shard_bounds_cover_the_grid_exactly():
    for ngrid in (1, 15, 16, 17, 1000, 143556, 1436406):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_bounds(ngrid, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == ngrid
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and a <= b
            assert all(lo % 16 == 0 for lo, hi in blocks if hi > lo)   # non-empty blocks start on a sub-tile
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, xc_type, ngrid, nao, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dm, ao, gr, w = synth_inputs(ngrid, nao, seed=77)      # same inputs on every rank
        lo, hi = shard_bounds(ngrid, world, rank)

        def local_sweep(dm_t):
            if hi <= lo:
                return 0.0, torch.zeros((nao, nao), dtype=torch.float64)
            e, v = oracle.compute_xc(xc_type, dm_t.numpy(), ao[lo:hi], w[lo:hi],
                                     None if xc_type == 0 else np.ascontiguousarray(gr[:, lo:hi]))
            return e, torch.from_numpy(v)

        sx = ShardedXC(nao, local_sweep, torch.device("cpu"))
        res = sx.compute_xc(torch.from_numpy(dm))
        np.save(os.path.join(out_dir, f"v{rank}.npy"), res.vxc.numpy())
        np.save(os.path.join(out_dir, f"e{rank}.npy"), np.array([res.exc]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("xc_type,ngrid,nao", [(1, 1003, 9), (2, 40, 5), (0, 17, 3)])
def test_two_rank_gloo_matches_unsharded(tmp_path, xc_type, ngrid, nao):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), xc_type, ngrid, nao, str(tmp_path)), nprocs=world, join=True)
    dm, ao, gr, w = synth_inputs(ngrid, nao, seed=77)
    e_ref, v_ref = oracle.compute_xc(xc_type, dm, ao, w, gr)
    for r in range(world):
        v = np.load(tmp_path / f"v{r}.npy"); e = float(np.load(tmp_path / f"e{r}.npy")[0])
        assert e == pytest.approx(e_ref, rel=1e-13)
        assert np.abs(v - v_ref).max() <= 1e-12 * np.abs(v_ref).max()
    assert np.array_equal(np.load(tmp_path / "v0.npy"), np.load(tmp_path / "v1.npy"))   # replicas agree bitwise


def test_vector_bounds_cover_the_vectors_exactly():
    from quantum_compute_dft_amd.grid_shard import vector_bounds
    for naux in (1, 7, 181, 3000):
        for world in (1, 2, 3, 8):
            blocks = [vector_bounds(naux, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == naux
            assert all(b == c for (_, b), (c, _) in zip(blocks, blocks[1:]))


def _fock_worker(rank, world, port, ngrid, nao, naux, nocc, out_dir):
    from quantum_compute_dft_amd.grid_shard import ShardedFock, vector_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, ao, gr, w = synth_inputs(ngrid, nao, seed=78)
        chol, cocc = _factors(nao, naux, nocc)
        dm = cocc @ cocc.T
        lo, hi = shard_bounds(ngrid, world, rank)
        plo, phi = vector_bounds(naux, world, rank)

        def local_sweep(dm_t):
            e, v = oracle.compute_xc(2, dm_t.numpy(), ao[lo:hi], w[lo:hi], np.ascontiguousarray(gr[:, lo:hi]))
            return e, torch.from_numpy(v)

        def local_jk(dm_t, cocc_t):
            J, K = oracle.jk_from_factors(chol[plo:phi], dm_t.numpy())
            return torch.from_numpy(J), torch.from_numpy(K)

        sf = ShardedFock(nao, local_sweep, local_jk, torch.device("cpu"))
        res = sf.compute(torch.from_numpy(dm), torch.from_numpy(cocc))
        np.savez(os.path.join(out_dir, f"f{rank}.npz"), v=res.vxc.numpy(), J=res.J.numpy(), K=res.K.numpy(), e=res.exc)
    finally:
        dist.destroy_process_group()


def _factors(nao, naux, nocc):
    rng = np.random.default_rng(5)
    A = rng.normal(0, 0.3, (naux, nao, nao))
    return 0.5 * (A + A.transpose(0, 2, 1)), rng.normal(0, 0.6, (nao, nocc))


def test_two_rank_gloo_fock_parts_match_unsharded(tmp_path):
    """Grid block + Cholesky-vector slice per rank, one all-reduce of [Vxc | J | K | Exc]."""
    world, ngrid, nao, naux, nocc = 2, 211, 7, 13, 3
    mp.spawn(_fock_worker, args=(world, _free_port(), ngrid, nao, naux, nocc, str(tmp_path)), nprocs=world, join=True)
    _, ao, gr, w = synth_inputs(ngrid, nao, seed=78)
    chol, cocc = _factors(nao, naux, nocc)
    dm = cocc @ cocc.T
    e_ref, v_ref = oracle.compute_xc(2, dm, ao, w, gr)
    J_ref, K_ref = oracle.jk_from_factors(chol, dm)
    r0, r1 = np.load(tmp_path / "f0.npz"), np.load(tmp_path / "f1.npz")
    for k in ("v", "J", "K"):
        assert np.array_equal(r0[k], r1[k])
    assert float(r0["e"]) == pytest.approx(e_ref, rel=1e-13)
    for got, ref in ((r0["v"], v_ref), (r0["J"], J_ref), (r0["K"], K_ref)):
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_eri_row_bounds_cover_whole_first_indices():
    from quantum_compute_dft_amd.grid_shard import eri_row_bounds
    for nao in (1, 7, 24, 114):
        for world in (1, 2, 3, 8, 200):
            blocks = [eri_row_bounds(nao, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == nao * nao
            assert all(b == c for (_, b), (c, _) in zip(blocks, blocks[1:]))
            assert all(lo % nao == 0 and hi % nao == 0 for lo, hi in blocks)      # whole i's: a rank's K rows are complete


def _eri_worker(rank, world, port, n, out_dir):
    """Row-sharded dense-ERI J/K under gloo: each rank contracts its row block (numpy restatement of what
    DFT_ComputeJKRows does on the device), one all-reduce of [J | K]."""
    from quantum_compute_dft_amd.grid_shard import eri_row_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(9)
        eri = rng.standard_normal((n * n, n * n)); dm = rng.standard_normal((n, n))
        lo, hi = eri_row_bounds(n, world, rank)
        J = (eri[lo:hi].T @ dm.ravel()[lo:hi]).reshape(n, n)                       # dft_solver.cu:550-555 on a row slice
        K = np.zeros((n, n))
        if hi > lo:
            blk = eri[lo:hi].reshape((hi - lo) // n, n, n, n)                      # (i, j, k, l), i in the block
            K[lo // n:hi // n] = np.einsum("ijkl,jl->ik", blk, dm)                 # dft.py:218 for these i
        buf = torch.from_numpy(np.concatenate([J.ravel(), K.ravel()]))
        dist.all_reduce(buf)
        np.save(os.path.join(out_dir, f"jk{rank}.npy"), buf.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_row_sharded_dense_jk(tmp_path):
    world, n = 2, 7
    mp.spawn(_eri_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(9)
    eri = rng.standard_normal((n * n, n * n)); dm = rng.standard_normal((n, n))
    J_ref, K_ref = oracle.coulomb(eri, dm), oracle.exchange(eri, dm)
    a, b = np.load(tmp_path / "jk0.npy"), np.load(tmp_path / "jk1.npy")
    assert np.array_equal(a, b)
    assert np.abs(a[: n * n].reshape(n, n) - J_ref).max() <= 1e-12 * np.abs(J_ref).max()
    assert np.abs(a[n * n:].reshape(n, n) - K_ref).max() <= 1e-12 * np.abs(K_ref).max()


def _chol_rank(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from quantum_compute_dft_amd import inputs
        inp = inputs.build("H2O", "def2-svp", 1, verbose=False, eri_mode="cholesky", chol_tol=1e-10, rank=rank, world=world)
        lo, hi, naux = inp.chol_range
        assert inp.chol.shape == (hi - lo, inp.shells.nao, inp.shells.nao)
        np.savez(os.path.join(out_dir, f"chol{rank}.npz"), L=np.asarray(inp.chol), lo=lo, hi=hi, naux=naux)
    finally:
        dist.destroy_process_group()


def test_cholesky_vectors_are_factorised_once_and_scattered(tmp_path):
    """inputs.build(world = 2): rank 0 alone factorises the ERI, each rank receives its vector slice
    (grid_shard.scatter_vectors) -- the slices put together are the single-rank vectors to 1e-12 (the reference builds its
    dense ERI once in its single process, grid.py:65; no rank here repeats the factorisation on 1/N of the cores)."""
    from quantum_compute_dft_amd import inputs
    from quantum_compute_dft_amd.grid_shard import vector_bounds
    world = 2
    mp.spawn(_chol_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = inputs.build("H2O", "def2-svp", 1, verbose=False, eri_mode="cholesky", chol_tol=1e-10)
    parts = [np.load(tmp_path / f"chol{r}.npz") for r in range(world)]
    naux = ref.chol.shape[0]
    assert all(int(p["naux"]) == naux for p in parts)
    for r, p in enumerate(parts):
        assert (int(p["lo"]), int(p["hi"])) == vector_bounds(naux, world, r)
    got = np.concatenate([p["L"] for p in parts])
    assert got.shape == ref.chol.shape
    assert np.abs(got - ref.chol).max() <= 1e-12
