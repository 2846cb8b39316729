// Launch side of the one-pass small-basis sweep kernel (fifth translation unit of libdft.so).
//
// Built with -mllvm -disable-machine-licm.  The kernel evaluates the functional inside its sub-tile loop, and the
// machine-level loop-invariant code motion of hipcc 7.2 hoists the functional's fp64 literals (two VGPRs each, about a
// hundred registers for B3LYP) in front of the loop, where they stay live across the matrix and plane registers:
// 127 / 187 / 221 VGPRs (LDA / GGA / B3LYP at nao <= 16) and 50 spilled registers for B3LYP at nao <= 32, against
// 78 / 140 / 135 and 197 without it (-Rpass-analysis=kernel-resource-usage).  The flag is per translation unit, hence
// this file.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "xc_tiny_kernels.hpp"
#include "xc_tiny_launch.hpp"

namespace qcdft {

static_assert(TINY_MAX_NAO == TN_MAX_NAO, "host and kernel limits");

// ONE workgroup per CU, the sub-tile loop takes the rest.  The registers would let 3 / 2 workgroups of the LDA variants
// share a CU (78 / 118 VGPRs; 512 threads are two waves on every SIMD), but a second workgroup does not start beside the
// first: the 13 extra workgroups of a 269-workgroup launch began 11.6 us after the others, when those had finished
// (tools/tiny_phase_probe.hip, profiles/r03_tiny_phase_probe.txt) -- a second dispatch round that pays the staging of
// the density matrix again and starts without the loop's prefetch.  Capped at one per CU the LDA kernel at 17-32
// functions went from 1.05-1.15 to 0.91-1.01 of the four launches between 33 k and 56 k points; the variants at
// <= 16 functions measure the same either way (profiles/r03_tiny_scan_one_per_cu.txt against r03_tiny_scan.txt).
int tiny_workgroups(int num_cu, int /*type*/, int /*nao*/, long ngrid)
{
    const long ntile = (ngrid + 15) / 16;
    return (int)std::max<long>(1, std::min<long>(num_cu, (ntile + TN_WAVES - 1) / TN_WAVES));
}

// Where the one-pass kernel is the faster call (tools/tiny_time.py scan / band on the final kernel,
// profiles/r03_tiny_scan_final.txt, r03_tiny_band_final.txt): everywhere it applies -- 0.71-0.93 of the four launches at
// nao <= 16, 0.75-1.00 at 17-32 functions, 20 k to 300 k points, all three functionals.  (Until a partial last round of
// sub-tiles was dealt one per CU instead of eight to the first workgroups, the sizes right above one sub-tile per wave,
// 32.8-42.6 k points, lost by up to 6 % and the rule excluded them: r03_tiny_band.txt.)
bool tiny_pays(int /*num_cu*/, int /*type*/, int nao, long /*ngrid*/)
{
    return nao <= TN_MAX_NAO;
}

void launch_sweep_tiny(hipStream_t st, int nwg, int type, long ngrid, int nao, const double *ao, const double *gx,
                       const double *gy, const double *gz, const double *dm, const double *w, double *slabs,
                       double *partial, int quirks)
{
    const dim3 g((unsigned)nwg), b(TN_THREADS);
#define QCDFT_TINY(T, S)                                                                                                     \
    if (nao <= 16) hipLaunchKernelGGL((k_sweep_tiny<1, T, S>), g, b, 0, st, ngrid, nao, ao, gx, gy, gz, dm, w, slabs, partial, quirks); \
    else           hipLaunchKernelGGL((k_sweep_tiny<2, T, S>), g, b, 0, st, ngrid, nao, ao, gx, gy, gz, dm, w, slabs, partial, quirks);
    if (type == 0)      { QCDFT_TINY(0, false) }
    else if (type == 1) { QCDFT_TINY(1, false) }
    else                { QCDFT_TINY(2, true) }
#undef QCDFT_TINY
}

} // namespace qcdft
