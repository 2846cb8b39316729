// Launch side of the occupied-orbital density kernels (second translation unit of libdft.so).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

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
    // all of C in LDS (row-shared kernel) when two workgroups still share a CU (160 KB of LDS); with ONE workgroup per CU
    // (QCDFT_OCC_RES_KB=150 at nao 246, 100 KB of C) it loses to the streamed kernel: 590 against 461 us
    size_t res_limit = 80 * 1024;
    if (const char *e = getenv("QCDFT_OCC_RES_KB")) res_limit = (size_t)atoi(e) * 1024;   // tools/occ_time.py
    p.resident = p.npass == 1 && p.nto <= 4 && occ_rs_lds_bytes(p.nto, p.nch) <= res_limit;
    // streamed kernels: four-wave workgroups (two per CU) up to 64 orbitals per pass -- eight waves sharing one stream of C
    // chunks halve its L2 traffic but march in lockstep: 961 against 928 us at the Anthracene/def2-TZVP shape
    p.nw = (p.resident || p.nto <= 4) ? 4 : 8;
    if (const char *e = getenv("QCDFT_OCC_NW")) p.nw = p.resident ? 4 : (atoi(e) == 4 && p.nto <= 4 ? 4 : 8);   // tools/occ_time.py
    p.lds_bytes = p.resident ? occ_rs_lds_bytes(p.nto, p.nch) : occ_lds_bytes(p.nto, p.nw, false, p.nch, p.npass);
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
                      const double *gy, const double *gz, const double *cp, double *rho, double *grad, double *sigma,
                      const double *cocc, int nocc)
{
    const long nrb = (ngrid + 16L * p.nw - 1) / (16L * p.nw);
    if constexpr (NTO <= 4) {
        if (p.resident) { // persistent row-shared workgroups: as many as the LDS lets live on the chip at once
            long cap = 3;   // three four-wave workgroups per CU where registers (<= 168) and LDS allow
            if (const char *e = getenv("QCDFT_OCC_PER_CU")) cap = std::max(1, atoi(e));   // tools/occ_time.py
            const long per_cu = std::max<long>(1, std::min<long>(cap, (160 * 1024) / (long)p.lds_bytes));
            const long ntile = (ngrid + 15) / 16;
            const unsigned grid = (unsigned)std::min<long>(ntile, per_cu * num_cu);
            auto launch = [&](auto kern, size_t &allowed) -> hipError_t {
                if (p.lds_bytes > allowed) {
                    const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
                    if (e != hipSuccess) return e;
                    allowed = p.lds_bytes;
                }
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), p.lds_bytes, st, ngrid, nao, p.nch, ao, gx, gy, gz, cocc, nocc, rho, grad, sigma);
                return hipGetLastError();
            };
            static size_t allowed1 = 48 * 1024, allowed4 = 48 * 1024;
            if (p.nch > 4) return launch(k_rho_occ_rs<NTO, GRAD, VEC, true>, allowed4);   // more than one chunk per wave
            return launch(k_rho_occ_rs<NTO, GRAD, VEC, false>, allowed1);
        }
        if (p.nw == 4) return launch_one<NTO, 4, GRAD, VEC, false, 1>(st, (unsigned)nrb, p.lds_bytes, ngrid, nao, p.nch, p.npass, ao, gx, gy, gz, cp, rho, grad, sigma);
        return launch_one<NTO, 8, GRAD, VEC, false, 1>(st, (unsigned)nrb, p.lds_bytes, ngrid, nao, p.nch, p.npass, ao, gx, gy, gz, cp, rho, grad, sigma);
    } else {
        return launch_one<NTO, 8, GRAD, VEC, false, 1>(st, (unsigned)nrb, p.lds_bytes, ngrid, nao, p.nch, p.npass, ao, gx, gy, gz, cp, rho, grad, sigma);
    }
}

template <bool GRAD, bool VEC>
hipError_t launch_gv(hipStream_t st, int num_cu, const OccPlan &p, long ngrid, int nao, const double *ao, const double *gx,
                     const double *gy, const double *gz, const double *cp, double *rho, double *grad, double *sigma,
                     const double *cocc, int nocc)
{
#define QCDFT_OCC(N) case N: return launch_nto<N, GRAD, VEC>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp, rho, grad, sigma, cocc, nocc)
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
    if (!p.resident) { // the streamed kernels read C in packed, zero-padded chunks; the row-shared one pads in its prologue
        const long n = (long)p.cp_doubles;
        hipLaunchKernelGGL(k_pack_cocc_occ, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, nao, nocc, 16 * p.nto, p.nch, p.npass, cocc, cp_ws);
    }
    if (gga) return vec16 ? launch_gv<true, true>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp_ws, rho, grad, sigma, cocc, nocc)
                          : launch_gv<true, false>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp_ws, rho, grad, sigma, cocc, nocc);
    return vec16 ? launch_gv<false, true>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp_ws, rho, grad, sigma, cocc, nocc)
                 : launch_gv<false, false>(st, num_cu, p, ngrid, nao, ao, gx, gy, gz, cp_ws, rho, grad, sigma, cocc, nocc);
}

void launch_dm_from_cocc(hipStream_t st, int nao, int nocc, const double *cocc, double *dm)
{
    const long n2 = (long)nao * nao;
    hipLaunchKernelGGL(k_dm_from_cocc, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, nao, nocc, cocc, dm);
}

} // namespace qcdft
