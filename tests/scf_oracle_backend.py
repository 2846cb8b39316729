"""CPU backend for scf.run_scf built on the oracle (TEST INFRASTRUCTURE: lives under tests/)."""
import time

import numpy as np

import oracle


class OracleBackend:
    def __init__(self, inp, functional, quirks=True):
        self.inp, self.t, self.q = inp, {"LDA": 0, "GGA": 1, "B3LYP": 2}[functional.upper()], quirks
        self.ao, self.gr = oracle.eval_ao(inp.shells, inp.grids.coords, deriv=1)

    def set_dm(self, dm):
        self.dm = np.ascontiguousarray(dm)

    def jk(self, want_k):
        return oracle.coulomb(self.inp.eri, self.dm), (oracle.exchange(self.inp.eri, self.dm) if want_k else None)

    def xc(self):
        t0 = time.time()
        e, v = oracle.compute_xc(self.t, self.dm, self.ao, self.inp.grids.weights, self.gr, quirks=self.q)
        return e, v, time.time() - t0
