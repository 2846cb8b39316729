"""Regenerate tests/golden/appendix_d_inputs.npz (inputs only; see the .cpp)."""
import os, subprocess, tempfile
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
with tempfile.TemporaryDirectory() as td:
    exe = os.path.join(td, "gen"); out = os.path.join(td, "in.bin")
    subprocess.run(["g++", "-O2", os.path.join(here, "gen_appendix_d_inputs.cpp"), "-o", exe], check=True)
    subprocess.run([exe, out], check=True)
    raw = np.fromfile(out, dtype="<f8")
ng, nao = 96, 5
o = 0
ao = raw[o:o + ng * nao].reshape(ng, nao); o += ng * nao
gr = raw[o:o + 3 * ng * nao].reshape(3, ng, nao); o += 3 * ng * nao
w = raw[o:o + ng]; o += ng
dm = raw[o:o + nao * nao].reshape(nao, nao); o += nao * nao
assert o == raw.size
np.savez(os.path.join(here, "appendix_d_inputs.npz"), ao=ao, ao_grad=gr, weights=w, dm=dm)
print("wrote appendix_d_inputs.npz")
