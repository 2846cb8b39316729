"""The SCF loop (dft.py:181-266 contract) on the CPU oracle backend: plumbing of inputs -> loop."""
import numpy as np
import pytest

from quantum_compute_dft_amd import inputs, scf
from scf_oracle_backend import OracleBackend


@pytest.fixture(scope="module")
def water():
    return inputs.build("H2O", "sto-3g", 3, verbose=False)


def test_inputs_are_consistent(water):
    assert water.shells.nao == 7 and water.grids.size == 34310 and water.nocc == 5
    assert np.allclose(water.S, water.S.T) and np.allclose(np.diag(water.S), 1.0, atol=1e-12)
    e = water.eri
    assert np.allclose(e, e.transpose(2, 3, 0, 1), atol=1e-12)


@pytest.mark.parametrize("fn,lo,hi", [("LDA", -74.80, -74.68), ("GGA", -75.30, -75.17), ("B3LYP", -75.39, -75.25)])
def test_scf_converges_and_counts_electrons(water, fn, lo, hi):
    be = OracleBackend(water, fn, quirks=False)
    r = scf.run_scf(water, be, fn, log=None)
    assert r["converged"] and r["cycles"] < 30
    assert lo < r["E_tot"] < hi                      # literature range for water / STO-3G
    rho = np.einsum("gi,ij,gj->g", be.ao, r["dm"], be.ao)
    assert float(water.grids.weights @ rho) == pytest.approx(10.0, abs=2e-4)   # integral of rho = N_elec
    assert np.trace(r["dm"] @ water.S) == pytest.approx(10.0, abs=1e-9)


def test_reference_derivative_quirks_shift_converged_energies(water):
    # measured: 4e-8 Ha (LDA, VWN5 dec_dx) and 4.5e-6 Ha (GGA, PBE-c dx_drho) on water/STO-3G
    for fn, lo, hi in (("LDA", 1e-9, 1e-5), ("GGA", 1e-7, 1e-4)):
        a = scf.run_scf(water, OracleBackend(water, fn, quirks=True), fn, log=None, conv_e=1e-11, conv_dm=1e-9)["E_tot"]
        b = scf.run_scf(water, OracleBackend(water, fn, quirks=False), fn, log=None, conv_e=1e-11, conv_dm=1e-9)["E_tot"]
        assert lo < abs(a - b) < hi, (fn, a - b)


def _drift_worker(rank, world, port, out_dir):
    """Two replicas of the host SCF loop under gloo; rank 1's LOCAL view of the device results is off by
    an ulp-scale perturbation every cycle (what different BLAS thread counts or different GPUs behind the
    eigensolver would do to unsynchronised replicas)."""
    import os
    import torch.distributed as dist
    from quantum_compute_dft_amd.grid_shard import ReplicaSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inp = inputs.build("H2O", "sto-3g", 1, verbose=False)

        class Drifting(OracleBackend):
            def xc(self):
                e, v, t = super().xc()
                if rank:
                    v = v * (1.0 + 3e-16) + 1e-17            # ~1 ulp off rank 0's matrix
                    e = e * (1.0 + 2e-16)
                return e, v, t

        be = Drifting(inp, "GGA")
        be.rank, be.world, be.replica_sync = rank, world, ReplicaSync("cpu")
        res = scf.run_scf(inp, be, "GGA", log=None, conv_e=1e-10, conv_dm=1e-8)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), dm=res["dm"], cycles=res["cycles"], conv=res["converged"],
                 e1=res["E_one"], ec=res["E_coul"])
    finally:
        dist.destroy_process_group()


def test_replicas_stay_bitwise_identical_and_stop_in_the_same_cycle(tmp_path):
    """rank 0 is authoritative (DIIS + eigh once, one broadcast of [dm | cocc | scalars] per cycle);
    a replica whose local numbers drift by an ulp still ends every cycle with rank 0's density, bit for bit,
    and leaves the loop in the same cycle (nobody is left alone in a collective)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_drift_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert bool(r0["conv"]) and bool(r1["conv"])
    assert int(r0["cycles"]) == int(r1["cycles"])
    assert np.array_equal(r0["dm"], r1["dm"])
    assert float(r0["e1"]) == float(r1["e1"]) and float(r0["ec"]) == float(r1["ec"])
    # and the synchronised run is the single-rank run
    inp = inputs.build("H2O", "sto-3g", 1, verbose=False)
    ref = scf.run_scf(inp, OracleBackend(inp, "GGA"), "GGA", log=None, conv_e=1e-10, conv_dm=1e-8)
    assert int(r0["cycles"]) == ref["cycles"] and np.array_equal(r0["dm"], ref["dm"])
