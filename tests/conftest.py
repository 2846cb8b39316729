import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def native_artefacts():
    """Every native artefact the tests load, built once per session when missing or stale (content hash);
    on the GPU box the prebuilt files of the snapshot match and nothing compiles.  Product entry points
    (bench.py, dft.py, smoke()) and the oracle loader never compile by themselves."""
    import oracle
    import quantum_compute_dft_amd as q
    from quantum_compute_dft_amd import integrals
    q.build_library()
    integrals.build_integrals()
    oracle.build(omp=False)
    oracle.build(omp=True)
