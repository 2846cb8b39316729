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

template <int NT, int TYPE, bool SYM> static int resident_per_cu()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_sweep_tiny<NT, TYPE, SYM>, TN_THREADS, 0) != hipSuccess || n < 1) n = 1;
    return n;
}

// Every workgroup of the launch is resident at once (a second dispatch round would pay the staging of the density
// matrix again and start without the prefetch of the sub-tile loop): the occupancy of the variant decides -- 512-thread
// workgroups put two waves on each SIMD, so 3 / 2 / 1 / 1 workgroups per CU at 78 / 118 / 135-140 / 197-201 VGPRs.
int tiny_workgroups(int num_cu, int type, int nao, long ngrid)
{
    static int per_cu[2][3] = {{0, 0, 0}, {0, 0, 0}};
    const int nt = nao <= 16 ? 0 : 1, ty = type < 0 || type > 2 ? 2 : type;
    if (!per_cu[nt][ty])
        per_cu[nt][ty] = nt == 0 ? (ty == 0 ? resident_per_cu<1, 0, false>() : ty == 1 ? resident_per_cu<1, 1, false>() : resident_per_cu<1, 2, true>())
                                 : (ty == 0 ? resident_per_cu<2, 0, false>() : ty == 1 ? resident_per_cu<2, 1, false>() : resident_per_cu<2, 2, true>());
    const long ntile = (ngrid + 15) / 16;
    const long cap = (long)std::min(per_cu[nt][ty], 2) * num_cu;
    return (int)std::max<long>(1, std::min<long>(cap, (ntile + TN_WAVES - 1) / TN_WAVES));
}

// Where the one-pass kernel is the faster call (tools/tiny_time.py scan / band, profiles/r03_tiny_scan.txt,
// r03_tiny_band.txt).  One column tile (nao <= 16): at every grid size measured (0.73-0.93 of the four launches).
// Two column tiles: the kernel's time is a staircase in rounds of R = 16 points x 8 waves x n_CU (one sub-tile per
// wave of one workgroup per CU; 32 768 points on 256 CUs) -- 0.80-0.85 at exactly R, 1.00-1.06 at R + 232, where the
// second sub-tile of a few waves costs a whole second chain -- while the four launches grow smoothly.  GGA / B3LYP (one
// workgroup per CU) lose by at most 4 % between R and 1.3 R and win by 2-17 % from there on (0.83-0.86 at 2 R, 0.86-0.92
// at 3.7 R, 0.79-0.93 at 9 R); LDA (two workgroups per CU, which slow each other) loses by 0-15 % up to 2 R and wins
// from 2.4 R.
bool tiny_pays(int num_cu, int type, int nao, long ngrid)
{
    if (nao > TN_MAX_NAO) return false;
    if (nao <= 16) return true;
    const double R = (double)num_cu * TN_WAVES * 16;
    return (double)ngrid <= R || (double)ngrid >= (type == 0 ? 2.2 : 1.3) * R;
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
