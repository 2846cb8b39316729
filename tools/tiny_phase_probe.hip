// Where the time of k_sweep_tiny's single-sub-tile chain goes: 100 MHz stamps left by every wave at its phase boundaries
// (QCDFT_TINY_STAMPS in csrc/xc_tiny_kernels.hpp), H2O/def2-SVP shape.  Build and run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -disable-machine-licm \
//         -Iquantum_compute_dft_amd/csrc tools/tiny_phase_probe.hip -o tools/tiny_phase_probe && tools/tiny_phase_probe
#define QCDFT_TINY_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "xc_tiny_kernels.hpp"
using namespace qcdft;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int TYPE> static int run(const char *name, long ngrid, int nao, int nwg)
{
    const size_t pl = (size_t)ngrid * nao;
    std::vector<double> h(pl * 4), hw(ngrid, 0.01), hd((size_t)nao * nao);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = 0.4 * ((double)(s >> 8) / (1 << 24) - 0.5); }
    for (int i = 0; i < nao; ++i) for (int j = 0; j < nao; ++j) hd[(size_t)i * nao + j] = (i == j ? 1.0 : 0.05) ;
    double *ao, *w, *dm, *slabs, *partial; unsigned long long *st;
    CK(hipMalloc(&ao, pl * 4 * 8)); CK(hipMalloc(&w, ngrid * 8)); CK(hipMalloc(&dm, hd.size() * 8));
    CK(hipMalloc(&slabs, (size_t)nwg * nao * nao * 8)); CK(hipMalloc(&partial, nwg * 8));
    const size_t nst = (size_t)nwg * TN_WAVES * 8;
    CK(hipMalloc(&st, nst * 8)); CK(hipMemset(st, 0, nst * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tiny_stamps), &st, sizeof(st)));
    CK(hipMemcpy(ao, h.data(), pl * 4 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), ngrid * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
    for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL((k_sweep_tiny<2, TYPE, false>), dim3(nwg), dim3(TN_THREADS), 0, 0, ngrid, nao, ao, ao + pl, ao + 2 * pl, ao + 3 * pl, dm, w, slabs, partial, 1);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> hs(nst);
    CK(hipMemcpy(hs.data(), st, nst * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t7 = 0;
    for (size_t wv = 0; wv < nst / 8; ++wv) if (hs[wv * 8]) { t0 = std::min(t0, hs[wv * 8]); t7 = std::max(t7, hs[wv * 8 + 7]); }
    static const char *ph[8] = {"start", "Ds staged + barrier", "planes landed", "X + row sums", "functional", "V MFMAs", "epilogue barriers", "slab stored"};
    printf("%s: ngrid %ld nao %d, %d workgroups; first start -> last end %.2f us (100 MHz stamps)\n", name, ngrid, nao, nwg, (t7 - t0) * 0.01);
    for (int i = 0; i < 8; ++i) {
        double sum = 0, mx = 0, mn = 1e30; size_t n = 0;
        for (size_t wv = 0; wv < nst / 8; ++wv) {
            if (!hs[wv * 8] || !hs[wv * 8 + i]) continue;
            const double t = (double)(hs[wv * 8 + i] - t0) * 0.01;
            sum += t; mx = std::max(mx, t); mn = std::min(mn, t); ++n;
        }
        printf("  %-22s reached at  min %6.2f  mean %6.2f  max %6.2f us after the first wave's start (%zu waves)\n", ph[i], mn, sum / std::max<size_t>(n, 1), mx, n);
    }
    hipFree(ao); hipFree(w); hipFree(dm); hipFree(slabs); hipFree(partial); hipFree(st);
    return 0;
}

int main()
{
    if (run<0>("LDA", 34310, 24, 269)) return 1;
    if (run<1>("GGA", 34310, 24, 256)) return 1;
    if (run<1>("GGA one round", 32768, 24, 256)) return 1;
    return 0;
}
