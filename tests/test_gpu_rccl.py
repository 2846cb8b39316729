"""RCCL on the hardware: a fresh child process initialises a torch.distributed "nccl" (= RCCL) group of ONE rank on cuda:0
and pushes the SCF's collective payloads -- [Vxc | Exc], [Vxc | J | K | Exc] and the [dm | cocc | scalars] broadcast --
through grid_shard.ShardedXC / ShardedFock / ReplicaSync with the collectives forced on (a one-rank group: the library
is loaded, the communicator is created, the kernels run; the 8-GPU node is the driver's to launch).  The reference has no
collective (single GPU: src/dft_solver.cu, dft.py)."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as dist
import quantum_compute_dft_amd as q
from quantum_compute_dft_amd.grid_shard import ReplicaSync, ShardedFock, ShardedXC
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
g = torch.Generator(device=dev); g.manual_seed(3)
ngrid, nao, nocc = 6000, 60, 11
ao = 0.4 * torch.randn((ngrid, nao), dtype=torch.float64, device=dev, generator=g)
gr = 0.3 * torch.randn((3, ngrid, nao), dtype=torch.float64, device=dev, generator=g)
w = 0.05 * torch.rand((ngrid,), dtype=torch.float64, device=dev, generator=g)
c = torch.randn((nao, nocc), dtype=torch.float64, device=dev, generator=g)
dm = (c @ c.T).contiguous()
s = q.DFTSolverWrapper(q.library_path(), "B3LYP")
v = torch.zeros((nao, nao), dtype=torch.float64, device=dev)
def sweep(d):
    e = s.compute_xc_occ(ngrid, nao, nocc, c, ao, w, v, gr, d)
    return e, v
e_ref, v_ref = sweep(dm); v_ref = v_ref.clone()
r = ShardedXC(nao, sweep, dev, collect_always=True).compute_xc(dm)
out["xc_exc_equal"] = bool(r.exc == e_ref); out["xc_vxc_equal"] = bool(torch.equal(r.vxc, v_ref))
J = torch.randn((nao, nao), dtype=torch.float64, device=dev, generator=g); K = torch.randn((nao, nao), dtype=torch.float64, device=dev, generator=g)
f = ShardedFock(nao, sweep, lambda d, cc: (J, K), dev, collect_always=True).compute(dm, c)
out["fock_equal"] = bool(f.exc == e_ref and torch.equal(f.vxc, v_ref) and torch.equal(f.J, J) and torch.equal(f.K, K))
out["fock_payload_bytes"] = 8 * (3 * nao * nao + 1)
sc = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64, device=dev)
d2, c2, s2 = dm.clone(), c.clone(), sc.clone()
ReplicaSync(dev, collect_always=True).broadcast([d2, c2, s2])
out["bcast_equal"] = bool(torch.equal(d2, dm) and torch.equal(c2, c) and torch.equal(s2, sc))
h = np.arange(12.0).reshape(3, 4).copy(); h0 = h.copy()
ReplicaSync(dev, collect_always=True).broadcast_numpy([h])
out["bcast_numpy_equal"] = bool(np.array_equal(h, h0))
t = torch.ones(1 << 20, dtype=torch.float64, device=dev)
dist.all_reduce(t); torch.cuda.synchronize()
out["allreduce_8MB_ok"] = bool(float(t.sum().item()) == float(1 << 20))
dist.barrier(); dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_rccl_group_of_one_rank_carries_the_scf_payloads(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    script = tmp_path / "rccl_child.py"
    script.write_text(CHILD)
    p = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert out["backend"] == "nccl" and out["world"] == 1
    for k in ("xc_exc_equal", "xc_vxc_equal", "fock_equal", "bcast_equal", "bcast_numpy_equal", "allreduce_8MB_ok"):
        assert out[k], (k, out)
