import time, numpy as np
from scipy.linalg import eigh
from threadpoolctl import threadpool_limits
def med(f, n=15):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
for n in (24, 114, 246, 494):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)); F = 0.5 * (A + A.T)
    B = rng.normal(size=(n, n)) * 0.3 / np.sqrt(n); S = np.eye(n) + 0.5 * (B + B.T)
    s, U = np.linalg.eigh(S); X = U / np.sqrt(s)
    out = []
    for th in (1, 2, 4, 8, 16):
        with threadpool_limits(limits=th):
            out.append((th, med(lambda: eigh(F, S)), med(lambda: np.linalg.eigh(X.T @ F @ X)), med(lambda: eigh(X.T @ F @ X, driver="evr")), med(lambda: eigh(X.T @ F @ X, driver="evd"))))
    print(f"n={n}: " + " | ".join(f"{th}t gvd {a:.2f} syevd(np) {b:.2f} evr {c:.2f} evd {d:.2f}" for th, a, b, c, d in out), flush=True)
