// Launch side of the occupied-orbital density kernels (second translation unit of libdft.so).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "xc_occ_kernels.hpp"
#include "xc_occ_launch.hpp"

namespace qcdft {

OccPlan occ_plan(int nao, int nocc, bool gga)
{
    OccPlan p;
    const int ntiles = (nocc + 15) / 16;
    p.npass = (ntiles + 7) / 8;
    p.nto = (ntiles + p.npass - 1) / p.npass;
    p.nch = (nao + OC_KC - 1) / OC_KC;
    // all of C in LDS when two four-wave workgroups still share a CU (160 KB of LDS)
    p.resident = p.nto <= 4 && occ_lds_bytes(p.nto, 4, true, p.nch, p.npass) <= 80 * 1024;
    p.nw = (p.resident || p.nto <= 4) ? 4 : 8;
    p.lds_bytes = occ_lds_bytes(p.nto, p.nw, p.resident, p.nch, p.npass);
    p.cp_doubles = (size_t)p.npass * p.nch * OC_KC * 16 * p.nto;
    const int NT = (nao + 15) / 16;
    p.mfma_occ = (double)p.npass * p.nch * 8.0 * p.nto * (gga ? 2.0 : 1.0);
    p.mfma_full = 4.0 * NT * NT;
    return p;
}

namespace {

template <int NTO, int NW, bool GRAD, bool VEC, bool RES, int GSETS>
hipError_t launch_one(hipStream_t st, unsigned grid, size_t lds, long ngrid, int nao, int nch, int npass, const double *ao,
                      const double *gx, const double *gy, const double *gz, const double *cp, double *rho, double *grad,
                      double *sigma)
{
    auto kern = k_rho_occ<NTO, NW, GRAD, VEC, RES, GSETS>;
    static size_t allowed = 48 * 1024; // per instantiation: dynamic LDS above the default needs the attribute once
    if (lds > allowed) {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        allowed = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, ngrid, nao, nch, npass, ao, gx, gy, gz, cp, rho, grad, sigma);
    return hipGetLastError();
}

template <int NTO, bool GRAD, bool VEC>
hipError_t launch_nto(hipStream_t st, int num_cu, const OccPlan &p, long ngrid, int nao, const double *ao, const double *gx,
                      const double *gy, const double *gz, const double *cp, double *rho, double *grad, double *sigma)
{
    const long nrb = (ngrid + 16L * p.nw - 1) / (16L * p.nw);
    if constexpr (NTO <= 4) {
        if (p.resident) { // persistent: as many workgroups as the LDS lets live on the chip at once
            const long per_cu = std::max<long>(1, std::min<long>(2, (160 * 1024) / (long)p.lds_bytes));
            const unsigned grid = (unsigned)std::min<long>(nrb, per_cu * num_cu);
            return launch_one<NTO, 4, GRAD, VEC, true, 1>(st, grid, p.lds_bytes, ngrid, nao, p.nch, p.npass, ao, gx, gy, gz, cp, rho, grad, sigma);
        }
        return launch_one<NTO, 4, GRAD, VEC, false, 1>(st, (unsigned)nrb, p.lds_bytes, ngrid, nao, p.nch, p.npass, ao, gx, gy, gz, cp, rho, grad, sigma);
    } else {
        return launch_one<NTO, 8, GRAD, VEC, false, 1>(st, (unsigned)nrb, p.lds_bytes, ngrid, nao, p.nch, p.npass, ao, gx, gy, gz, cp, rho, grad, sigma);
    }
}

template <bool GRAD, bool VEC>
hipError_t launch_gv(hipStream_t st, int num_cu, const OccPlan &p, long ngrid, int nao, const double *ao, const double *gx,
                     const double *gy, const double *gz, const double *cp, double *rho, double *grad, double *sigma)
{
#define QCDFT_OCC(N) case N: return launch_nto<N, GRAD, VEC>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp, rho, grad, sigma)
    switch (p.nto) {
        QCDFT_OCC(1); QCDFT_OCC(2); QCDFT_OCC(3); QCDFT_OCC(4);
        QCDFT_OCC(5); QCDFT_OCC(6); QCDFT_OCC(7); QCDFT_OCC(8);
    default: return hipErrorInvalidValue;
    }
#undef QCDFT_OCC
}

} // namespace

hipError_t launch_rho_occ(hipStream_t st, int num_cu, const OccPlan &p, bool gga, bool vec16, long ngrid, int nao, int nocc,
                          const double *cocc, double *cp_ws, const double *ao, const double *gx, const double *gy,
                          const double *gz, double *rho, double *grad, double *sigma)
{
    const long n = (long)p.cp_doubles;
    hipLaunchKernelGGL(k_pack_cocc_occ, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, nao, nocc, 16 * p.nto, p.nch, p.npass, cocc, cp_ws);
    if (gga) return vec16 ? launch_gv<true, true>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp_ws, rho, grad, sigma)
                          : launch_gv<true, false>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp_ws, rho, grad, sigma);
    return vec16 ? launch_gv<false, true>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp_ws, rho, grad, sigma)
                 : launch_gv<false, false>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp_ws, rho, grad, sigma);
}

void launch_dm_from_cocc(hipStream_t st, int nao, int nocc, const double *cocc, double *dm)
{
    const long n2 = (long)nao * nao;
    hipLaunchKernelGGL(k_dm_from_cocc, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, nao, nocc, cocc, dm);
}

} // namespace qcdft
