// Where do the ~98 us of k_rho_occ_rs (Benzene/def2-SVP shape) go?  The kernel as shipped against ablated builds of the
// same source (results wrong by construction): -DNO_MFMA (matrix instructions replaced by one add), -DNO_BARRIER
// (the two workgroup barriers per tile dropped).  HBM floor of this access pattern: 85 us (stream_pattern_probe3, D).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -w [-DNO_MFMA] [-DNO_BARRIER] \
//        -Iquantum_compute_dft_amd/csrc -o tools/occ_ablate[_x] tools/occ_ablate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "device_util.hpp"
#ifdef NO_MFMA
namespace qcdft { __device__ __forceinline__ d4 fake_mfma(double a, double b, d4 c) { c[0] += a + b; return c; } }
#define mfma_f64 fake_mfma
#endif
#ifdef NO_BARRIER
#define lds_barrier() __builtin_amdgcn_wave_barrier()
#endif
#ifdef NO_STORE
#define QCDFT_OCC_NO_STORE 1
#endif
#ifdef NO_COMPUTE
#define QCDFT_OCC_ABL_NOCOMPUTE 1
#endif
#ifdef NO_EXCHANGE
#define QCDFT_OCC_ABL_NOEXCHANGE 1
#endif
#ifdef NO_REDUCE
#define QCDFT_OCC_ABL_NOREDUCE 1
#endif
#include "xc_occ_kernels.hpp"
using namespace qcdft;
#ifndef ABL
#define ABL ""
#endif
int main(int argc, char **argv)
{
    const long ngrid = 143556; const int nao = 114, nocc = 21, nto = 2, nch = 4;
    const int per_cu = argc > 1 ? atoi(argv[1]) : 3;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); const int ncu = prop.multiProcessorCount;
    const size_t plane = (size_t)ngrid * nao;
    double *p, *cp, *rho, *grad, *sigma;
    hipMalloc(&p, plane * 8 * 4 + 4096); hipMalloc(&cp, 8 * nao * nocc); hipMalloc(&rho, 8 * ngrid); hipMalloc(&grad, 24 * ngrid); hipMalloc(&sigma, 8 * ngrid);
    std::vector<double> h(plane * 4); srand(1); for (auto &x : h) x = (rand() / (double)RAND_MAX - 0.5) * 0.8;
    hipMemcpy(p, h.data(), plane * 8 * 4, hipMemcpyHostToDevice);
    std::vector<double> c(nao * nocc, 0.0); for (auto &x : c) x = (rand() / (double)RAND_MAX - 0.5);
    hipMemcpy(cp, c.data(), c.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = occ_rs_lds_bytes(nto, nch);
    auto kern = k_rho_occ_rs<2, true, true, false>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);

    const unsigned grid = (unsigned)(per_cu * ncu);
    auto L = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, ngrid, nao, nch, p, p + plane, p + 2 * plane, p + 3 * plane, cp, nocc, rho, grad, sigma); };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 300; ++r) L();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 100; ++r) L();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 100;
    printf("k_rho_occ_rs<2,GGA> " ABL " %s%s per_cu=%d lds=%zu: %.1f us  %.0f GB/s\n",
#ifdef NO_MFMA
           "NO_MFMA ",
#else
           "",
#endif
#ifdef NO_BARRIER
           "NO_BARRIER",
#else
           "",
#endif
           per_cu, lds, ms * 1e3, 4.0 * plane * 8 / ms * 1e-6);
    return 0;
}
