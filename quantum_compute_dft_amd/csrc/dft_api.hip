// C-ABI of libdft.so (include/dft_solver.h) and the host orchestration behind it.
//
// Replaces src/dft_solver.cu:530-719 of the reference (XCSolver classes, the
// per-call cudaMalloc/cudaFree sequence and the four extern "C" entry points).
// Differences by design: one persistent workspace per solver instead of 6-7
// allocations per call; 64-bit index arithmetic; deterministic reductions;
// errors are recorded (DFT_GetLastError) as well as printed.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <limits>
#include <string>
#include <vector>

#include "../../include/dft_solver.h"
#include "ao_kernels.hpp"
#include "jk_kernels.hpp"
#include "cd_kernels.hpp"
#include "xc_big_kernels.hpp"
#include "xc_ws_kernels.hpp"
#include "xc_ws16_kernels.hpp"
#include "xc_kernels.hpp"
#include "xc_occ_launch.hpp"
#include "xc_tiny_launch.hpp"

using namespace qcdft;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Timing {
    const char *name;
    hipEvent_t t0, t1;
};

// One recorded sweep (option "graph"): the launches of a call with exactly these arguments, replayed as one HIP graph.
struct SweepGraph {
    long ngrid = 0;
    int nao = 0, nocc = 0;
    const void *dm = nullptr, *ao = nullptr, *grad = nullptr, *w = nullptr, *vxc = nullptr, *cocc = nullptr;
    hipGraphExec_t exec = nullptr;
    int seen = 0;      // calls with this key so far (the first sizes the workspace, the second is recorded)
    bool bad = false;  // recording failed once: this key runs as plain launches from then on
    unsigned long stamp = 0;
};

} // namespace

struct XCSolver {
    int type = 0;
    hipStream_t stream = nullptr;
    bool device_ok = false;
    int device = 0; // the device that was current at DFT_CreateSolver: every entry point runs there
    int num_cu = 256;
    // options
    int quirks = 1;
    int path = 0; // 0 auto (wave-specialised persistent kernels when nao <= 128), 1 VALU validation, 2 generic MFMA
    int profile = 0;
    int ksplit = 0;
    int ao_pt = 0; // grid points per workgroup of the AO kernel: 0 auto, 8 or 16
    // 0 (default): Exc is finished by a one-block kernel of its own BEHIND the Vxc reduce -- stream order puts it after
    // every store of the call, so the host-mapped word means "the whole call has completed" for any consumer (other
    // streams, mapped host reads): the reference's contract (blocking copy + cudaFree, dft_solver.cu:575-582).
    // 1 (opt-in): the reduce kernel's highest-index block finishes Exc itself (one launch and ~1.5 us fewer); the word
    // then only orders consumers on the solver's own stream.
    int fuse_finish = 0;
    int rho_rows = 64; // grid rows per workgroup of the large-basis rho kernel: 64 (two workgroups per CU) or 128
    int sweep_order = 2; // bit 0: rho kernel walks the grid backwards, bit 1: Vxc kernel does (default: rho forward, Vxc backward)
    int dbg = 0;       // diagnostics only (ablations of the sixteen-wave kernels: 1 = no plane loads, 2 = no MFMAs)
    int tiny = -1;     // one-pass sweep kernel for nao <= 32 (xc_tiny_kernels.hpp): -1 auto (where it is faster, tiny_pays()), 0 off, 1 on
    int ws_waves = 0;  // wave-specialised kernels (nao <= 128): 0 auto, 8 = 4+4 waves per workgroup, 16 = 8+8
    int occ = 0;       // DFT_ComputeXCOcc: 0 auto (occupied-orbital density step where it does fewer MFMAs), 1 always, 2 never
    int used_occ = 0;  // what the last sweep did (DFT_GetTimings names say so too)
    int eri_sym = 0;   // 1: the caller vouches that the dense ERI is symmetric as an (N2, N2) matrix: DFT_ComputeCoulomb streams its upper
                       // triangle only; 2: ... and in each index pair, and dm = dm^T: the unique eighth only
    // A synchronous call seen before with the same pointers and sizes is replayed as one recorded HIP graph (one submission
    // instead of five launches): -1 auto = where the call is launch-bound (planes of at most GRAPH_AUTO_ELEMS doubles: H2O/def2-SVP
    // 30.4 -> 26.3 us per LDA call, 34.7 -> 32.9 GGA; Benzene/STO-3G 84.5 -> 86.0 and Benzene/def2-SVP 228.4 -> 230.6, so not there),
    // 1 always, 0 never.  An option change and any growth of the workspace drop the recorded graphs.
    int graph = -1;
    std::vector<SweepGraph> graphs;
    hipStream_t cap_stream = nullptr; // recording happens here (the caller's stream may be the null stream, which cannot record)
    unsigned long graph_clock = 0, graph_gen = 0;
    // workspace
    DevBuf dsym, rho, sigma, grad, coef, partial, slabs, exc, jpart, kpart, shells, msym, cdy, cdc, cdv, ao_ws, vtmp, occ_cp, occ_dm;
    int spin_wait = 1; // poll the host-mapped Exc instead of sleeping in hipStreamSynchronize
    int strict_sync = 0; // 1: after the Exc word, also poll the stream until it reports complete (+8-10 us per call)
    double *h_exc = nullptr;   // pinned, host-mapped: the reduce kernel writes Exc here
    double *h_exc_dev = nullptr; // device alias of h_exc
    std::string last_error;
    std::vector<Timing> timings;
    size_t n_timed = 0;
    // AO shell-table cache key
    std::vector<unsigned char> shell_blob;
};

namespace {

void set_error(XCSolver *s, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fprintf(stderr, "libdft: %s\n", buf);
    if (s) s->last_error = buf;
}

bool hip_ok(XCSolver *s, hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    set_error(s, "%s failed: %s", what, hipGetErrorString(e));
    return false;
}

void drop_graphs(XCSolver *s)
{
    for (SweepGraph &g : s->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    s->graphs.clear();
    ++s->graph_gen;
}

bool reserve(XCSolver *s, DevBuf &b, size_t bytes, const char *what)
{
    if (bytes <= b.cap) return true;
    drop_graphs(s); // recorded launches hold the old workspace pointers
    if (b.p) {
        (void)hipStreamSynchronize(s->stream);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    if (!hip_ok(s, hipMalloc(&b.p, want), what)) return false;
    b.cap = want;
    return true;
}

// Every entry point runs on the solver's device (its workspace and stream live there): a caller that
// switched devices since DFT_CreateSolver gets the solver's device for the call and its own back after.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const XCSolver *s)
    {
        if (!s || !s->device_ok) return;
        if (hipGetDevice(&prev) == hipSuccess && prev != s->device) switched = hipSetDevice(s->device) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

struct ScopedTimer {
    XCSolver *s;
    size_t idx = 0;
    bool on;
    ScopedTimer(XCSolver *s_, const char *name) : s(s_), on(s_->profile != 0)
    {
        if (!on) return;
        idx = s->n_timed++;
        if (idx >= s->timings.size()) {
            Timing t{name, nullptr, nullptr};
            (void)hipEventCreate(&t.t0);
            (void)hipEventCreate(&t.t1);
            s->timings.push_back(t);
        }
        s->timings[idx].name = name;
        (void)hipEventRecord(s->timings[idx].t0, s->stream);
    }
    ~ScopedTimer()
    {
        if (on) (void)hipEventRecord(s->timings[idx].t1, s->stream);
    }
};

int auto_ksplit(const XCSolver *s, long ngrid, int nblk)
{
    if (s->ksplit > 0) return s->ksplit;
    // enough workgroups to fill the chip a few times over, >= 256 points each
    long want = (4L * s->num_cu + nblk - 1) / nblk;
    long maxsplit = (ngrid + 255) / 256;
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    return (int)want;
}

#define QCDFT_NT_SWITCH(NTV, CALL)              \
    switch (NTV) {                               \
    case 1: { constexpr int NT = 1; CALL; } break; \
    case 2: { constexpr int NT = 2; CALL; } break; \
    case 3: { constexpr int NT = 3; CALL; } break; \
    case 4: { constexpr int NT = 4; CALL; } break; \
    case 5: { constexpr int NT = 5; CALL; } break; \
    case 6: { constexpr int NT = 6; CALL; } break; \
    case 7: { constexpr int NT = 7; CALL; } break; \
    default: { constexpr int NT = 8; CALL; } break; \
    }

// The sweep: everything on s->stream, Exc left in s->exc (device).
// `cocc` (nao, nocc) with dm = cocc cocc^T switches the density step to the occupied-orbital form where that
// pays (xc_occ_kernels.hpp); `dm` may then be null.
bool xc_sweep(XCSolver *s, long ngrid, int nao, const double *dm, const double *ao,
              const double *ao_grad, const double *w, double *vxc, bool want_host_exc,
              const double *cocc = nullptr, int nocc = 0, double *exc_out = nullptr)
{
    s->last_error.clear();
    s->n_timed = 0;
    if (!s->device_ok) {
        set_error(s, "no usable HIP device");
        return false;
    }
    if (ngrid <= 0 || nao <= 0) {
        set_error(s, "bad sizes ngrid=%ld nao=%d", ngrid, nao);
        return false;
    }
    const bool gga = s->type != SOLVER_LDA;
    if (gga && !ao_grad) {
        set_error(s, "ao_grad pointer is null for a gradient-corrected functional");
        return false;
    }
    if (!dm && !cocc) {
        set_error(s, "neither a density matrix nor occupied orbitals were given");
        return false;
    }
    if (cocc && nocc <= 0) {
        set_error(s, "occupied orbitals given with nocc=%d", nocc);
        return false;
    }
    // Occupied-orbital density step: 16 nch nocc_tiles MFMAs per 16 grid rows (8 for LDA) against 4 NT^2 through the
    // full matrix; taken when it does clearly fewer (the two kernels run at similar matrix-pipe efficiency), on the
    // production path only.
    // Small bases: the whole sweep in one kernel (planes below 4 GiB: one buffer descriptor each)
    const bool tiny = s->path == 0 && s->tiny != 0 && nao <= TINY_MAX_NAO && (double)ngrid * nao * 8.0 < 4294967296.0 &&
                      (s->tiny > 0 || tiny_pays(s->num_cu, s->type == SOLVER_LDA ? 0 : s->type == SOLVER_GGA ? 1 : 2, nao, ngrid));
    OccPlan oplan;
    bool use_occ = false;
    if (cocc && s->path == 0 && s->occ != 2 && !tiny) {
        oplan = occ_plan(nao, nocc, gga);
        use_occ = s->occ == 1 || oplan.mfma_occ <= 0.85 * oplan.mfma_full;
    }
    s->used_occ = use_occ;
    if (!use_occ && !dm) { // the dm kernels need the matrix itself
        if (!reserve(s, s->occ_dm, sizeof(double) * (size_t)nao * nao, "hipMalloc(dm)")) return false;
        launch_dm_from_cocc(s->stream, nao, nocc, cocc, (double *)s->occ_dm.p);
        dm = (const double *)s->occ_dm.p;
    }
    const int NP = ((nao + 15) / 16) * 16;
    const int nblk = (nao + 127) / 128;
    // the wave-specialised kernels address a plane through ONE buffer descriptor: planes of 4 GiB or more
    // (ngrid*nao >= 2^29) take the generic tiled kernels
    const bool fits32 = (double)ngrid * nao * 8.0 < 4294967296.0;
    const bool fast = s->path == 0 && nao <= 128 && fits32;
    const bool big = s->path == 0 && nao > 128;   // nao <= 128 with planes >= 4 GiB: the generic MFMA kernels below
    const int ntv = NP / 16;
    const bool ws16 = fast && s->ws_waves == 16; // opt-in: measured equal to the eight-wave kernels at NT = 8, slower below
    int nslab;
    long chunk = 0;
    const int nA = (nao + BG_BM - 1) / BG_BM, nB = (nao + BG_BN - 1) / BG_BN, npair = nA * nB;
    if (big) {
        // split-K over grid chunks; chunks are dealt to XCDs (blockIdx % 8), so ksplit is a multiple of 8
        // chunks per XCD: the candidate (<= 32, slabs <= 2 GB, >= 64 grid rows per chunk) whose workgroup
        // count fills whole waves of the chip best (360 workgroups on 256 CUs lost 30 % to the tail)
        long per_xcd = 1;
        double best = 0.0;
        for (long c = 1; c <= 32; ++c) {
            const long wgs = 8L * npair * c;
            if (8 * c * 64 > ngrid && c > 1) break;
            if ((double)(8 * c) * nao * nao * 8.0 > 2.0e9 && c > 1) break;
            const long rounds = (wgs + s->num_cu - 1) / s->num_cu;
            const double eff = (double)wgs / (double)(rounds * s->num_cu);
            if (eff > best + 1e-9) { best = eff; per_xcd = c; }
        }
        // a chunk of one plane stays below 4 GiB: k_vxc_big addresses it through one buffer descriptor
        while ((double)((ngrid + 8 * per_xcd - 1) / (8 * per_xcd) + BG_BK) * nao * 8.0 >= 4294967296.0) ++per_xcd;
        nslab = (int)(8 * per_xcd);
        chunk = (ngrid + nslab - 1) / nslab;
        chunk = ((chunk + BG_BK - 1) / BG_BK) * BG_BK;
    } else if (tiny) {
        nslab = tiny_workgroups(s->num_cu, s->type == SOLVER_LDA ? 0 : s->type == SOLVER_GGA ? 1 : 2, nao, ngrid);
    } else if (fast) {
        const long ntile = (ngrid + WS_ROWS - 1) / WS_ROWS;
        nslab = (int)std::min<long>(s->num_cu, ntile); // one persistent, wave-specialised workgroup per CU
    } else {
        const int nsplit = auto_ksplit(s, ngrid, nblk * nblk);
        chunk = (ngrid + nsplit - 1) / nsplit;
        chunk = ((chunk + 31) / 32) * 32;
        nslab = (int)((ngrid + chunk - 1) / chunk);
    }
    const long nxb = tiny ? nslab : (ngrid + 255) / 256; // Exc partials: one per workgroup of the kernel that evaluates the functional
    const size_t ng = (size_t)ngrid;

    if (!reserve(s, s->dsym, sizeof(double) * NP * NP, "hipMalloc(Dsym)") ||
        !reserve(s, s->rho, sizeof(double) * ng, "hipMalloc(rho)") ||
        !reserve(s, s->coef, sizeof(double) * ng * (gga ? 4 : 1), "hipMalloc(coef)") ||
        !reserve(s, s->partial, sizeof(double) * nxb, "hipMalloc(partial)") ||
        !reserve(s, s->slabs, sizeof(double) * (size_t)nslab * nao * nao, "hipMalloc(slabs)") ||
        false)
        return false;
    if (!reserve(s, s->exc, 2 * sizeof(double), "hipMalloc(exc)")) return false; // the device-side Exc scalar
    if (gga && (!reserve(s, s->sigma, sizeof(double) * ng, "hipMalloc(sigma)") ||
                !reserve(s, s->grad, sizeof(double) * 3 * ng, "hipMalloc(grad)")))
        return false;

    double *Dp = (double *)s->dsym.p, *rho = (double *)s->rho.p, *sigma = (double *)s->sigma.p,
           *grad = (double *)s->grad.p, *coef = (double *)s->coef.p,
           *partial = (double *)s->partial.p, *slabs = (double *)s->slabs.p,
           *exc = exc_out ? exc_out : (double *)s->exc.p;   // the asynchronous entries' caller-owned scalar: written by the finishing kernel itself
    const double *gx = ao_grad, *gy = gga ? ao_grad + ng * nao : nullptr,
                 *gz = gga ? ao_grad + 2 * ng * nao : nullptr;
    hipStream_t st = s->stream;

    if (tiny) {
        ScopedTimer t(s, "sweep_tiny");
        launch_sweep_tiny(st, nslab, s->type == SOLVER_LDA ? 0 : s->type == SOLVER_GGA ? 1 : 2, ngrid, nao, ao, gx, gy, gz, dm, w,
                          slabs, partial, s->quirks);
    }
    if (use_occ) {
        ScopedTimer t(s, "rho_occ");
        if (!reserve(s, s->occ_cp, sizeof(double) * oplan.cp_doubles, "hipMalloc(packed cocc)")) return false;
        const bool vec16 = (nao % 2 == 0) && ((((uintptr_t)ao | (uintptr_t)gx | (uintptr_t)gy | (uintptr_t)gz) & 15) == 0);
        if (!hip_ok(s, launch_rho_occ(st, s->num_cu, oplan, gga, vec16, ngrid, nao, nocc, cocc, (double *)s->occ_cp.p, ao, gx, gy, gz,
                                      rho, grad, sigma), "occupied-orbital density launch"))
            return false;
    }
    if (!fast && !use_occ && !tiny) { // the wave-specialised rho kernel symmetrises D in its prologue
        ScopedTimer t(s, "sym_dm");
        dim3 b(16, 16), g(NP / 16, NP / 16);
        hipLaunchKernelGGL(k_sym_dm, g, b, 0, st, nao, NP, dm, Dp);
    }
    if (!use_occ && !tiny) {
        ScopedTimer t(s, "rho");
        const int vec16 = (nao % 2 == 0) && ((((uintptr_t)ao | (uintptr_t)gx | (uintptr_t)gy | (uintptr_t)gz) & 15) == 0);
        if (fast && ws16) {
            dim3 g((unsigned)nslab);
#define QCDFT_RHO(G, V) QCDFT_NT_SWITCH(ntv, hipLaunchKernelGGL((k_rho_ws16<NT, G, V>), g, dim3(W16_THREADS), 0, st, ngrid, nao, ao, gx, gy, gz, dm, rho, grad, sigma, s->dbg | ((s->sweep_order & 1) << 16)))
            if (gga) { if (vec16) { QCDFT_RHO(true, true) } else { QCDFT_RHO(true, false) } }
            else     { if (vec16) { QCDFT_RHO(false, true) } else { QCDFT_RHO(false, false) } }
#undef QCDFT_RHO
        } else if (fast) {
            dim3 g((unsigned)nslab);
#define QCDFT_RHO(G, V) QCDFT_NT_SWITCH(ntv, hipLaunchKernelGGL((k_rho_ws<NT, G, V>), g, dim3(WS_THREADS), 0, st, ngrid, nao, ao, gx, gy, gz, dm, rho, grad, sigma, s->sweep_order & 1))
            if (gga) { if (vec16) { QCDFT_RHO(true, true) } else { QCDFT_RHO(true, false) } }
            else     { if (vec16) { QCDFT_RHO(false, true) } else { QCDFT_RHO(false, false) } }
#undef QCDFT_RHO
        } else if (big) {
            if (s->rho_rows == 128) {
                dim3 g((unsigned)((ngrid + BG_BM - 1) / BG_BM));
                if (gga) { if (vec16) hipLaunchKernelGGL((k_rho_big<true, true>), g, dim3(BG_THREADS), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma);
                           else       hipLaunchKernelGGL((k_rho_big<true, false>), g, dim3(BG_THREADS), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma); }
                else     { if (vec16) hipLaunchKernelGGL((k_rho_big<false, true>), g, dim3(BG_THREADS), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma);
                           else       hipLaunchKernelGGL((k_rho_big<false, false>), g, dim3(BG_THREADS), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma); }
            } else { // 64-row workgroups, two per CU
                dim3 g((unsigned)((ngrid + R6_BM - 1) / R6_BM));
                if (gga) { if (vec16) hipLaunchKernelGGL((k_rho_big64<true, true>), g, dim3(256), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma);
                           else       hipLaunchKernelGGL((k_rho_big64<true, false>), g, dim3(256), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma); }
                else     { if (vec16) hipLaunchKernelGGL((k_rho_big64<false, true>), g, dim3(256), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma);
                           else       hipLaunchKernelGGL((k_rho_big64<false, false>), g, dim3(256), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma); }
            }
        } else if (s->path != 1) {
            dim3 g((unsigned)((ngrid + 63) / 64));
            if (gga) hipLaunchKernelGGL(k_rho_mfma<true>, g, dim3(256), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma);
            else     hipLaunchKernelGGL(k_rho_mfma<false>, g, dim3(256), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma);
        } else {
            dim3 g((unsigned)((ngrid + 3) / 4));
            if (gga) hipLaunchKernelGGL(k_rho_valu<true>, g, dim3(256), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma);
            else     hipLaunchKernelGGL(k_rho_valu<false>, g, dim3(256), 0, st, ngrid, nao, NP, ao, gx, gy, gz, Dp, rho, grad, sigma);
        }
    }
    if (!tiny) {
        ScopedTimer t(s, "xc_points");
        dim3 g((unsigned)nxb);
        if (s->type == SOLVER_LDA)      hipLaunchKernelGGL(k_xc_points<0>, g, dim3(256), 0, st, ngrid, rho, sigma, grad, w, coef, partial, s->quirks);
        else if (s->type == SOLVER_GGA) hipLaunchKernelGGL(k_xc_points<1>, g, dim3(256), 0, st, ngrid, rho, sigma, grad, w, coef, partial, s->quirks);
        else                            hipLaunchKernelGGL(k_xc_points<2>, g, dim3(256), 0, st, ngrid, rho, sigma, grad, w, coef, partial, s->quirks);
    }
    if (!tiny) {
        ScopedTimer t(s, "vxc");
        const int vec16 = (nao % 2 == 0) && ((((uintptr_t)ao | (uintptr_t)gx | (uintptr_t)gy | (uintptr_t)gz) & 15) == 0);
        if (fast && ws16) {
            dim3 g((unsigned)nslab);
#define QCDFT_VXC(G, V, S) QCDFT_NT_SWITCH(ntv, hipLaunchKernelGGL((k_vxc_ws16<NT, G, V, S>), g, dim3(W16_THREADS), 0, st, ngrid, nao, ao, gx, gy, gz, coef, slabs, s->dbg | ((s->sweep_order & 2) << 15)))
            if (s->type == SOLVER_B3LYP) { if (vec16) { QCDFT_VXC(true, true, true) } else { QCDFT_VXC(true, false, true) } }
            else if (gga) { if (vec16) { QCDFT_VXC(true, true, false) } else { QCDFT_VXC(true, false, false) } }
            else          { if (vec16) { QCDFT_VXC(false, true, false) } else { QCDFT_VXC(false, false, false) } }
#undef QCDFT_VXC
        } else if (fast) {
            dim3 g((unsigned)nslab);
#define QCDFT_VXC(G, V, S) QCDFT_NT_SWITCH(ntv, hipLaunchKernelGGL((k_vxc_ws<NT, G, V, S>), g, dim3(WS_THREADS), 0, st, ngrid, nao, ao, gx, gy, gz, coef, slabs, (s->sweep_order >> 1) & 1))
            if (s->type == SOLVER_B3LYP) { if (vec16) { QCDFT_VXC(true, true, true) } else { QCDFT_VXC(true, false, true) } }
            else if (gga) { if (vec16) { QCDFT_VXC(true, true, false) } else { QCDFT_VXC(true, false, false) } }
            else          { if (vec16) { QCDFT_VXC(false, true, false) } else { QCDFT_VXC(false, false, false) } }
#undef QCDFT_VXC
        } else if (big) {
            dim3 g((unsigned)(nslab * npair));
            if (gga) { if (vec16) hipLaunchKernelGGL((k_vxc_big<true, true>), g, dim3(BG_THREADS), 0, st, ngrid, nao, chunk, nB, npair, ao, gx, gy, gz, coef, slabs);
                       else       hipLaunchKernelGGL((k_vxc_big<true, false>), g, dim3(BG_THREADS), 0, st, ngrid, nao, chunk, nB, npair, ao, gx, gy, gz, coef, slabs); }
            else     { if (vec16) hipLaunchKernelGGL((k_vxc_big<false, true>), g, dim3(BG_THREADS), 0, st, ngrid, nao, chunk, nB, npair, ao, gx, gy, gz, coef, slabs);
                       else       hipLaunchKernelGGL((k_vxc_big<false, false>), g, dim3(BG_THREADS), 0, st, ngrid, nao, chunk, nB, npair, ao, gx, gy, gz, coef, slabs); }
        } else if (s->path != 1) {
            dim3 g((unsigned)nslab, nblk, nblk);
            if (gga) hipLaunchKernelGGL(k_vxc_mfma<true>, g, dim3(256), 0, st, ngrid, nao, chunk, ao, gx, gy, gz, coef, slabs);
            else     hipLaunchKernelGGL(k_vxc_mfma<false>, g, dim3(256), 0, st, ngrid, nao, chunk, ao, gx, gy, gz, coef, slabs);
        } else {
            dim3 g((unsigned)nslab);
            if (gga) hipLaunchKernelGGL(k_vxc_valu<true>, g, dim3(256), 0, st, ngrid, nao, chunk, ao, gx, gy, gz, coef, slabs);
            else     hipLaunchKernelGGL(k_vxc_valu<false>, g, dim3(256), 0, st, ngrid, nao, chunk, ao, gx, gy, gz, coef, slabs);
        }
    }
    {
        ScopedTimer t(s, "reduce_vxc");
        dim3 g((unsigned)(((size_t)nao * nao + 31) / 32));
        const bool b3 = s->type == SOLVER_B3LYP;
        if (b3 && big) { // plain reduce into M, then tiled M + M^T
            if (!reserve(s, s->msym, sizeof(double) * (size_t)nao * nao, "hipMalloc(M)")) return false;
            double *M = (double *)s->msym.p;
            hipLaunchKernelGGL(k_reduce_slabs8<false>, g, dim3(256), 0, st, nao, nslab, slabs, M);
            const int nt = (nao + 31) / 32;
            hipLaunchKernelGGL(k_symmetrize, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, st, nao, M, vxc);
        } else if (b3 && !fast) {
            hipLaunchKernelGGL(k_reduce_slabs8<true>, g, dim3(256), 0, st, nao, nslab, slabs, vxc);
        } else if (s->fuse_finish) { // wave-specialised slabs are already symmetrised for B3LYP
            // last launch of the call: its last-ticket block also finishes Exc (device scalar + host-mapped word)
            hipLaunchKernelGGL((k_reduce_slabs8<false, true>), g, dim3(256), 0, st, nao, nslab, slabs, vxc, nxb, partial, exc,
                               want_host_exc ? s->h_exc_dev : nullptr);
            return hip_ok(s, hipGetLastError(), "XC sweep launch");
        } else {
            // Exc is summed by the reduce kernel's highest-index block into the DEVICE scalar only; publishing it to the host
            // is left to a one-thread kernel behind it (stream order: the whole call has completed when the word appears) --
            // the shortest launch there is, instead of a finishing kernel that still has the partials to add up
            hipLaunchKernelGGL((k_reduce_slabs8<false, true>), g, dim3(256), 0, st, nao, nslab, slabs, vxc, nxb, partial, exc, (double *)nullptr);
            if (want_host_exc && s->h_exc_dev) hipLaunchKernelGGL(k_publish_exc, dim3(1), dim3(1), 0, st, exc, s->h_exc_dev);
            return hip_ok(s, hipGetLastError(), "XC sweep launch");
        }
        // last launch of the call: Exc to the device scalar and to host-mapped memory
        hipLaunchKernelGGL(k_finish_exc, dim3(1), dim3(256), 0, st, nxb, partial, exc, want_host_exc ? s->h_exc_dev : nullptr);
    }
    return hip_ok(s, hipGetLastError(), "XC sweep launch");
}

// J and/or K from rows (i, j), i in [i0, i0 + ni), of the dense ERI; `eri` points at row (i0, 0).
// i0 = 0, ni = nao is the reference's whole-matrix contraction; a proper sub-range is one rank's share of
// the row sharding: partial J over all columns, rows [i0, i0 + ni) of K, zeros elsewhere.
void jk(XCSolver *s, int nao, const double *eri, const double *dm, double *J, double *K, int i0 = 0, int ni = -1)
{
    s->last_error.clear();
    if (!s->device_ok) { set_error(s, "no usable HIP device"); return; }
    if (nao <= 0 || nao > JK_COLS) { set_error(s, "dense-ERI J/K needs 1 <= nao <= %d (got %d)", JK_COLS, nao); return; }
    if (ni < 0) ni = nao;
    if (i0 < 0 || ni <= 0 || i0 + ni > nao) { set_error(s, "dense-ERI J/K: bad row range [%d, %d) of %d", i0, i0 + ni, nao); return; }
    if (!J && !K) return;
    const int n = nao;
    const size_t N2 = (size_t)n * n;
    const int KB = std::max(1, std::min(n, JK_COLS / n));
    const int ncb = (n + KB - 1) / KB;
    // (below ~48 functions the whole pass is a handful of workgroups either way and the symmetric kernels' extra slab sums
    //  cost more than their bytes save: H2O/def2-SVP 39 against 30 us)
    if (J && !K && s->eri_sym == 2 && i0 == 0 && ni == n && n >= 48) {   // the unique eighth only (k_j_sym8)
        const size_t NPK = (size_t)n * (n + 1) / 2;
        // packed columns per thread (4, 2, 1): the widest blocks that still leave more live workgroups than CUs (half of the
        // (block, chunk) pairs are live; measured: nao 114 -> 4 (52.9 us; 1: 67.2), nao 80 -> 1 (31.5 us; 4: 41.7))
        const int nchunk = (int)((NPK + JS8_RC - 1) / JS8_RC);
        int cpt = 4;
        while (cpt > 1 && (double)((NPK + 256 * cpt - 1) / (256 * cpt)) * nchunk / 2.0 < 1.2 * s->num_cu) cpt /= 2;
        {
            const char *ov = getenv("QCDFT_JSYM8_CPT");   // (tools/jsym_sweep.sh)
            if (ov && (atoi(ov) == 1 || atoi(ov) == 2 || atoi(ov) == 4)) cpt = atoi(ov);
        }
        const int ncb8 = (int)((NPK + 256 * cpt - 1) / (256 * cpt)), nslab = nchunk + ncb8;
        if (!reserve(s, s->jpart, sizeof(double) * (nslab + 1) * NPK, "hipMalloc(Jpart)")) return;
        double *jp = (double *)s->jpart.p;
        if (cpt == 4)      hipLaunchKernelGGL((k_j_sym8<4>), dim3(ncb8, nchunk), dim3(256), 0, s->stream, n, eri, dm, jp);
        else if (cpt == 2) hipLaunchKernelGGL((k_j_sym8<2>), dim3(ncb8, nchunk), dim3(256), 0, s->stream, n, eri, dm, jp);
        else               hipLaunchKernelGGL((k_j_sym8<1>), dim3(ncb8, nchunk), dim3(256), 0, s->stream, n, eri, dm, jp);
        hipLaunchKernelGGL(k_sum_slabs8_sym, dim3((unsigned)((NPK + 31) / 32)), dim3(256), 0, s->stream, n, nslab, jp, J);
        hip_ok(s, hipGetLastError(), "J launch");
        return;
    }
    if (J && !K && s->eri_sym && i0 == 0 && ni == n && n >= 48) {   // upper triangle only (k_j_sym)
        const int nslab = n + ncb;
        if (!reserve(s, s->jpart, sizeof(double) * nslab * N2, "hipMalloc(Jpart)")) return;
        double *jp = (double *)s->jpart.p;
        const bool vec = (n % 2 == 0) && (((uintptr_t)eri & 15) == 0);
        if (vec) hipLaunchKernelGGL((k_j_sym<true>), dim3(ncb, n), dim3(256), 0, s->stream, n, KB, eri, dm, jp);
        else     hipLaunchKernelGGL((k_j_sym<false>), dim3(ncb, n), dim3(256), 0, s->stream, n, KB, eri, dm, jp);
        hipLaunchKernelGGL(k_sum_slabs8, dim3((unsigned)((N2 + 31) / 32)), dim3(256), 0, s->stream, N2, nslab, N2, jp, J);
        hip_ok(s, hipGetLastError(), "J launch");
        return;
    }
    // enough workgroups to fill the chip: split the j range when ni*ncb is small
    int jsplit = 1;
    while ((long)ni * ncb * jsplit < 4L * s->num_cu && jsplit * 2 <= n) jsplit *= 2;
    const int nslabJ = ni * jsplit;
    if (J && !reserve(s, s->jpart, sizeof(double) * nslabJ * N2, "hipMalloc(Jpart)")) return;
    if (K && !reserve(s, s->kpart, sizeof(double) * jsplit * (size_t)ni * n, "hipMalloc(Kpart)")) return;
    double *jp = (double *)s->jpart.p, *kp = (double *)s->kpart.p;
    dim3 g(ncb, ni * jsplit);
    hipStream_t st = s->stream;
    const bool vec = (n % 2 == 0) && (((uintptr_t)eri & 15) == 0);
#define QCDFT_JK(WJ, WK)                                                                                                  \
    do {                                                                                                                  \
        if (vec) hipLaunchKernelGGL((k_jk_stream<WJ, WK, true>), g, dim3(256), 0, st, n, KB, jsplit, i0, ni, eri, dm, jp, kp);    \
        else     hipLaunchKernelGGL((k_jk_stream<WJ, WK, false>), g, dim3(256), 0, st, n, KB, jsplit, i0, ni, eri, dm, jp, kp);   \
    } while (0)
    if (J && K) QCDFT_JK(true, true);
    else if (J) QCDFT_JK(true, false);
    else        QCDFT_JK(false, true);
#undef QCDFT_JK
    if (J) hipLaunchKernelGGL(k_sum_slabs8, dim3((unsigned)((N2 + 31) / 32)), dim3(256), 0, st, N2, nslabJ, N2, jp, J);
    if (K) {
        const size_t nk = (size_t)ni * n;
        if (ni != n) (void)hipMemsetAsync(K, 0, sizeof(double) * N2, st);
        hipLaunchKernelGGL(k_sum_slabs8, dim3((unsigned)((nk + 31) / 32)), dim3(256), 0, st, nk, jsplit, nk, kp, K + (size_t)i0 * n);
    }
    hip_ok(s, hipGetLastError(), "J/K launch");
}


// chunks per XCD for a split-K launch of `npair` output tiles: the count (<= 32) whose workgroup
// total fills whole rounds of the chip best, with slabs <= 2 GB and >= 64 contraction rows per chunk
long chunks_per_xcd(const XCSolver *s, int npair, long rows, size_t slab_bytes)
{
    // the fewest chunks whose workgroup total fills at least 92 % of whole rounds of the chip (more
    // chunks mean more slabs to sum and shorter contraction loops), else the best filling found
    long best_c = 1;
    double best = 0.0;
    for (long c = 1; c <= 32; ++c) {
        const long wgs = 8L * npair * c;
        if (8 * c * 64 > rows && c > 1) break;
        if ((double)(8 * c) * (double)slab_bytes > 2.0e9 && c > 1) break;
        const long rounds = (wgs + s->num_cu - 1) / s->num_cu;
        const double eff = (double)wgs / (double)(rounds * s->num_cu);
        if (eff >= 0.92) return c;
        if (eff > best + 1e-9) { best = eff; best_c = c; }
    }
    return best_c;
}

// J and/or K from Cholesky vectors L (naux, nao, nao), D = dm, dm = cocc cocc^T with cocc (nao, nocc)
int jk_factorized(XCSolver *s, int nao, int naux, int nocc, const double *L, const double *dm,
                  const double *cocc, double *J, double *K)
{
    s->last_error.clear();
    if (!s->device_ok) { set_error(s, "no usable HIP device"); return -1; }
    if (nao <= 0 || naux <= 0 || !L) { set_error(s, "factorised J/K: bad sizes nao=%d naux=%d", nao, naux); return -1; }
    if (J && !dm) { set_error(s, "factorised J needs the density matrix"); return -1; }
    if (K && (!cocc || nocc <= 0)) { set_error(s, "factorised K needs occupied orbitals (nocc=%d)", nocc); return -1; }
    if (!J && !K) return 0;
    hipStream_t st = s->stream;
    const long n2 = (long)nao * nao;
    const int nB = (nao + CD_BN - 1) / CD_BN;
    // half transform: 64 x 256 tiles (8-row stages, four waves) while one 64-row tile holds all occupied orbitals,
    // 128 x 256 above; one partial of v_P per tile
    const int nBh = nB, npair_h = ((nocc + (nocc <= 64 ? 63 : 127)) / (nocc <= 64 ? 64 : 128)) * nBh;
    // v_P = L_P : D.  With K wanted too, the half transform leaves it as the dot of its result Yt_P with Cocc
    // (D = Cocc Cocc^T is the entry point's contract), one partial per tile; otherwise a pass of its own over L with D.
    const bool fused_dot = J && K;
    if (J) {
        if (!reserve(s, s->cdv, sizeof(double) * ((size_t)naux * (2 + npair_h) + 2), "hipMalloc(cd v)")) return -1;
        if (!fused_dot) {
            ScopedTimer t(s, "cd_dot");
            hipLaunchKernelGGL(k_cd_dot, dim3((unsigned)naux), dim3(256), 0, st, n2, L, dm, (double *)s->cdv.p);
        }
    }
    double *v = (double *)s->cdv.p, *vpart = v ? v + naux : nullptr;
    if (K) {
        const int ldp = ((nocc + 15) / 16) * 16;          // padded occupied dimension of the A operand
        const int ldy = (nao + 1) & ~1;                   // even leading dimension of Yt: 16-byte rows
        const long G = (long)naux * nocc;                 // rows of Yt
        const int nA2 = (nao + 127) / 128, npair2 = nA2 * nB;
        int live2 = 0; // tiles of K = Yt^T Yt that are computed (the rest are mirrored)
        for (int ia = 0; ia < nA2; ++ia)
            for (int ib = 0; ib < nB; ++ib) live2 += !(128 * ia >= CD_BN * ib + CD_BN);
        long per_xcd = s->ksplit > 0 ? s->ksplit : chunks_per_xcd(s, live2, G, sizeof(double) * (size_t)n2);
        while ((double)((G + 8 * per_xcd - 1) / (8 * per_xcd) + CD_BK) * ldy * 8.0 >= 4294967296.0) per_xcd *= 2; // a chunk of Yt stays below 4 GiB (one buffer descriptor)
        const int nslab = (int)(8 * per_xcd);
        long chunk = (G + nslab - 1) / nslab;
        chunk = ((chunk + CD_BK - 1) / CD_BK) * CD_BK;
        const bool fresh = sizeof(double) * (size_t)G * ldy > s->cdy.cap;
        if (!reserve(s, s->cdc, sizeof(double) * (size_t)nao * ldp, "hipMalloc(cd cocc)") ||
            !reserve(s, s->cdy, sizeof(double) * (size_t)G * ldy, "hipMalloc(cd Yt)") ||
            !reserve(s, s->kpart, sizeof(double) * (size_t)nslab * n2, "hipMalloc(cd K slabs)"))
            return -1;
        double *cp = (double *)s->cdc.p, *yt = (double *)s->cdy.p, *kp = (double *)s->kpart.p;
        if (fresh && ldy != nao) // the pad column is read (into discarded outputs) but never written
            if (!hip_ok(s, hipMemsetAsync(yt, 0, s->cdy.cap, st), "memset(cd Yt)")) return -1;
        const bool vecL = (nao % 2 == 0) && (((uintptr_t)L & 15) == 0);
        {
            ScopedTimer t(s, "cd_half");
            hipLaunchKernelGGL(k_pack_cocc, dim3((unsigned)(((long)nao * ldp + 255) / 256)), dim3(256), 0, st, nao, nocc, ldp, cocc, cp);
            // Yt_P (nocc x nao) = Cp^T L_P for every P
#define QCDFT_HALF3(WGM, MI, VL, DOT, NW)                                                                             \
    hipLaunchKernelGGL((k_gemm_tn<WGM, MI, true, VL, DOT, NW, (WGM == 1 ? 8 : 0), (WGM == 1 ? 4 : 0)>), g, dim3(64 * NW), 0, st, (long)nao, nocc, nao, ldp, nao, \
                       cp, 0L, L, n2, (long)nao, nBh, npair, 0, yt, ldy, (long)nocc * ldy, 0L, vpart)
#define QCDFT_HALF(WGM, MI, NW)                                                                                       \
    do {                                                                                                              \
        const int nA = (nocc + 64 * WGM - 1) / (64 * WGM), npair = nA * nBh;                                          \
        dim3 g((unsigned)((long)npair * naux));                                                                       \
        if (fused_dot) { if (vecL) QCDFT_HALF3(WGM, MI, true, true, NW); else QCDFT_HALF3(WGM, MI, false, true, NW); }   \
        else           { if (vecL) QCDFT_HALF3(WGM, MI, true, false, NW); else QCDFT_HALF3(WGM, MI, false, false, NW); } \
    } while (0)
            if (nocc <= 16) QCDFT_HALF(1, 1, 4);
            else if (nocc <= 32) QCDFT_HALF(1, 2, 4);
            else if (nocc <= 48) QCDFT_HALF(1, 3, 4);
            else if (nocc <= 64) QCDFT_HALF(1, 4, 4);
            else QCDFT_HALF(2, 4, 8);
#undef QCDFT_HALF
#undef QCDFT_HALF3
        }
        {
            ScopedTimer t(s, "cd_k");
            // K = Yt^T Yt, split over the (P, i) rows
            dim3 g((unsigned)(nslab * live2));
            hipLaunchKernelGGL((k_gemm_tn<2, 4, true, true>), g, dim3(BG_THREADS), 0, st, G, nao, nao, ldy, ldy,
                               yt, 0L, yt, 0L, chunk, nB, live2, 1, kp, nao, 0L, n2, (double *)nullptr, 1);
            hipLaunchKernelGGL(k_sum_slabs8, dim3((unsigned)((n2 + 31) / 32)), dim3(256), 0, st, (size_t)n2, nslab, (size_t)n2, kp, K);
            if (live2 < npair2)
                hipLaunchKernelGGL(k_mirror_lower, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, nao, K);
        }
    }
    if (J) {
        ScopedTimer t(s, "cd_j");
        // J = sum_P v_P L_P: slices of vectors so that the pass has >= ~4 workgroups per CU
        const long eb = (n2 + 255) / 256;
        int nsl = (int)std::max<long>(1, std::min<long>(naux, (4L * s->num_cu + eb - 1) / eb));
        const int pslice = (naux + nsl - 1) / nsl;
        nsl = (naux + pslice - 1) / pslice;
        if (!reserve(s, s->jpart, sizeof(double) * (size_t)nsl * n2, "hipMalloc(cd J slabs)")) return -1;
        double *jp = (double *)s->jpart.p;
        if (fused_dot) {
            // The fused dots are L_P : (cocc cocc^T).  A dm that is NOT that product (damped, mixed, fractional occupations)
            // must give ITS Coulomb matrix: checked on the device, and v_P = L_P : dm is then taken in a pass of its own
            // (a launch that exits at once when the two agree) -- no host round trip either way.
            double *vdot = vpart + (size_t)naux * npair_h;
            int *flag = (int *)(vdot + naux);
            (void)hipMemsetAsync(flag, 0, sizeof(int), st);
            hipLaunchKernelGGL(k_dm_consistency, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, nao, nocc, dm, cocc, flag);
            hipLaunchKernelGGL(k_cd_dot_if, dim3((unsigned)naux), dim3(256), 0, st, flag, n2, L, dm, vdot);
            hipLaunchKernelGGL(k_cd_vsum, dim3((unsigned)((naux + 255) / 256)), dim3(256), 0, st, naux, npair_h, vpart, v, flag, vdot);
        }
        hipLaunchKernelGGL(k_cd_axpy, dim3((unsigned)eb, (unsigned)nsl), dim3(256), 0, st, n2, naux, pslice, L, v, jp, nao);
        hipLaunchKernelGGL(k_sum_slabs8, dim3((unsigned)((n2 + 31) / 32)), dim3(256), 0, st, (size_t)n2, nsl, (size_t)n2, jp, J);
        hipLaunchKernelGGL(k_sym_from_upper, dim3((unsigned)eb), dim3(256), 0, st, nao, J);
    }
    return hip_ok(s, hipGetLastError(), "factorised J/K launch") ? 0 : -1;
}

// The AO shell table on the device: validated, packed and uploaded once per distinct table
// ([AoShell x nshell][exp x nprim][coef x nprim][AoChunk x nchunk][order x nshell]).
struct AoTable {
    const AoShell *sh; const double *exp, *coef; const AoChunk *chunks; const int *order;
    int nshell, nprim, nchunk, maxcol; bool even_blocks;
};

bool prepare_ao_table(XCSolver *s, int nao, int nshell, const double *shl_xyz, const int *shl_l, const int *shl_nprim,
                      const int *shl_off, const int *shl_ao, const double *prim_exp, const double *prim_coef, int nprim_total,
                      AoTable &out)
{
    if (nao <= 0 || nshell <= 0 || nprim_total <= 0) {
        set_error(s, "bad AO sizes");
        return false;
    }
    int next_col = 0;
    std::vector<AoChunk> chunks;
    for (int i = 0; i < nshell; ++i) {
        const int l = shl_l[i], nf = 2 * l + 1;
        if (l < 0 || l > AO_MAX_L) { set_error(s, "shell %d: l=%d unsupported (max %d)", i, l, AO_MAX_L); return false; }
        if (shl_nprim[i] <= 0 || shl_off[i] < 0 || shl_off[i] + shl_nprim[i] > nprim_total) { set_error(s, "shell %d: primitive range out of bounds", i); return false; }
        if (shl_ao[i] != next_col) { set_error(s, "shell %d: AO columns must be contiguous and ascending (expected %d, got %d)", i, next_col, shl_ao[i]); return false; }
        if (chunks.empty()) {
            chunks.push_back(AoChunk{i, i, next_col, 0});
        } else if (chunks.back().ncol + nf > AO_CW) {
            // new column block; keep its first column even (16-byte stores) by taking the previous
            // shell along when needed -- every shell has an odd width, so that flips the parity
            AoChunk &b = chunks.back();
            if ((next_col & 1) && b.shell_hi - b.shell_lo >= 2) {
                const int pf = 2 * shl_l[i - 1] + 1;
                b.shell_hi -= 1;
                b.ncol -= pf;
                chunks.push_back(AoChunk{i - 1, i, next_col - pf, pf});
            } else {
                chunks.push_back(AoChunk{i, i, next_col, 0});
            }
        }
        chunks.back().shell_hi = i + 1;
        chunks.back().ncol += nf;
        next_col += nf;
    }
    if (next_col != nao) { set_error(s, "shell table covers %d AO columns, nao=%d", next_col, nao); return false; }
    const int nchunk = (int)chunks.size();
    int maxcol = 0;
    bool even = true;
    std::vector<int> order(nshell);
    for (const AoChunk &c : chunks) {
        maxcol = std::max(maxcol, c.ncol);
        if (c.col_lo & 1) even = false;
        for (int i = c.shell_lo; i < c.shell_hi; ++i) order[i] = i;
        std::stable_sort(order.begin() + c.shell_lo, order.begin() + c.shell_hi, [&](int a, int b) {
            return shl_l[a] != shl_l[b] ? shl_l[a] < shl_l[b] : shl_nprim[a] < shl_nprim[b];
        });
    }
    const size_t off_exp = sizeof(AoShell) * nshell;
    const size_t off_chunk = off_exp + 2 * sizeof(double) * nprim_total;
    const size_t off_order = off_chunk + sizeof(AoChunk) * nchunk;
    const size_t bytes = off_order + sizeof(int) * nshell;
    std::vector<unsigned char> blob(bytes);
    AoShell *sh = (AoShell *)blob.data();
    for (int i = 0; i < nshell; ++i) {
        sh[i].x = shl_xyz[3 * i]; sh[i].y = shl_xyz[3 * i + 1]; sh[i].z = shl_xyz[3 * i + 2];
        sh[i].l = shl_l[i]; sh[i].nprim = shl_nprim[i]; sh[i].off = shl_off[i]; sh[i].ao = shl_ao[i];
    }
    double *pe = (double *)(blob.data() + off_exp);
    memcpy(pe, prim_exp, sizeof(double) * nprim_total);
    memcpy(pe + nprim_total, prim_coef, sizeof(double) * nprim_total);
    memcpy(blob.data() + off_chunk, chunks.data(), sizeof(AoChunk) * nchunk);
    memcpy(blob.data() + off_order, order.data(), sizeof(int) * nshell);
    if (blob != s->shell_blob) {
        if (!reserve(s, s->shells, bytes, "hipMalloc(shells)")) return false;
        if (!hip_ok(s, hipMemcpyAsync(s->shells.p, blob.data(), bytes, hipMemcpyHostToDevice, s->stream), "upload shells") ||
            !hip_ok(s, hipStreamSynchronize(s->stream), "synchronise"))
            return false;
        s->shell_blob.swap(blob);
    }
    const unsigned char *base = (const unsigned char *)s->shells.p;
    out.sh = (const AoShell *)base;
    out.exp = (const double *)(base + off_exp);
    out.coef = out.exp + nprim_total;
    out.chunks = (const AoChunk *)(base + off_chunk);
    out.order = (const int *)(base + off_order);
    out.nshell = nshell; out.nprim = nprim_total; out.nchunk = nchunk; out.maxcol = maxcol; out.even_blocks = even;
    return true;
}

void launch_ao(XCSolver *s, const AoTable &t, long ngrid, int nao, const double *coords, double *ao, double *ao_grad)
{
    const bool vec = t.even_blocks && (nao % 2 == 0) && ((uintptr_t)ao % 16 == 0) && ((uintptr_t)ao_grad % 16 == 0);
    // 16 points per workgroup unless their LDS tile would leave fewer than three workgroups per CU
    // (measured, Benzene: def2-SVP deriv 1 139 -> 122 us with 8; STO-3G and deriv 0 are 10 % faster with 16)
    const int ao_pt = s->ao_pt ? s->ao_pt : ((ao_grad ? 4 : 1) * 16 * (t.maxcol | 1) * 8 > 53 * 1024 ? 8 : 16);
    launch_eval_ao(s->stream, ngrid, nao, t.nchunk, t.maxcol, vec, ao_pt, s->num_cu, t.nshell, t.nprim, t.sh, t.exp, t.coef, t.chunks, t.order,
                   coords, ao, ao_grad);
}

// V (+)= V_chunk, Exc (+)= Exc_chunk: the grid chunks of the direct sweep add up by linearity in the grid points
__global__ void k_accumulate_chunk(long n2, int add, const double *__restrict__ vc, const double *__restrict__ ec,
                                   double *__restrict__ v, double *__restrict__ e)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) v[i] = add ? v[i] + vc[i] : vc[i];
    if (i == 0) *e = add ? *e + *ec : *ec;
}

} // namespace

extern "C" {

int DFT_GetVersion(void) { return 3; }

XCSolver *DFT_CreateSolver(int type)
{
    if (type != SOLVER_LDA && type != SOLVER_GGA && type != SOLVER_B3LYP) return nullptr;
    XCSolver *s = new (std::nothrow) XCSolver();
    if (!s) return nullptr;
    s->type = type;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            s->device_ok = true;
            s->device = dev;
            s->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        if (s->device_ok) {
            if (hipHostMalloc((void **)&s->h_exc, sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
                hipHostGetDevicePointer((void **)&s->h_exc_dev, s->h_exc, 0) != hipSuccess) {
                (void)hipGetLastError();
                s->h_exc = nullptr;
                s->h_exc_dev = nullptr;
            }
        }
    } else {
        (void)hipGetLastError();
        s->last_error = "no usable HIP device";
    }
    return s;
}

void DFT_DestroySolver(XCSolver *s)
{
    if (!s) return;
    if (s->device_ok) {
        DeviceGuard dg(s);
        (void)hipStreamSynchronize(s->stream);
        drop_graphs(s);
        if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
        DevBuf *bufs[] = {&s->dsym, &s->rho, &s->sigma, &s->grad, &s->coef, &s->partial,
                          &s->slabs, &s->exc, &s->jpart, &s->kpart, &s->shells, &s->msym,
                          &s->cdy, &s->cdc, &s->cdv, &s->ao_ws, &s->vtmp, &s->occ_cp, &s->occ_dm};
        for (DevBuf *b : bufs)
            if (b->p) (void)hipFree(b->p);
        if (s->h_exc) (void)hipHostFree(s->h_exc);
        for (Timing &t : s->timings) {
            (void)hipEventDestroy(t.t0);
            (void)hipEventDestroy(t.t1);
        }
    }
    delete s;
}

// Option "graph": the launches of a synchronous call, recorded once per (pointers, sizes) and replayed as one HIP graph.
constexpr double GRAPH_AUTO_ELEMS = 2.0e6;

struct SweepArgs {
    long ngrid;
    int nao, nocc;
    const double *dm, *ao, *grad, *w;
    double *vxc;
    const double *cocc;
};

static SweepGraph *find_sweep(XCSolver *s, const SweepArgs &a)
{
    for (SweepGraph &g : s->graphs)
        if (g.ngrid == a.ngrid && g.nao == a.nao && g.nocc == a.nocc && g.dm == a.dm && g.ao == a.ao && g.grad == a.grad &&
            g.w == a.w && g.vxc == a.vxc && g.cocc == a.cocc)
            return &g;
    return nullptr;
}

// Records the sweep for `a` on the solver's recording stream; on any failure the key is marked and runs as plain launches.
static bool record_sweep(XCSolver *s, const SweepArgs &a)
{
    if (!s->cap_stream && hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const unsigned long gen = s->graph_gen;
    if (hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    hipStream_t user = s->stream;
    s->stream = s->cap_stream;
    const bool ok = xc_sweep(s, a.ngrid, a.nao, a.dm, a.ao, a.grad, a.w, a.vxc, true, a.cocc, a.nocc);
    s->stream = user;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(s->cap_stream, &graph);
    hipGraphExec_t exec = nullptr;
    const bool good = ok && e == hipSuccess && graph && gen == s->graph_gen &&
                      hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!good) (void)hipGetLastError();
    SweepGraph *g = gen == s->graph_gen ? find_sweep(s, a) : nullptr;
    if (!g) {
        if (exec) (void)hipGraphExecDestroy(exec);
        return false;
    }
    g->exec = good ? exec : nullptr;
    g->bad = !good;
    return good;
}

// True when the call was submitted as a recorded graph (recording it first if this is the key's second call).
static bool replay_sweep(XCSolver *s, const SweepArgs &a)
{
    SweepGraph *g = find_sweep(s, a);
    if (!g || g->bad) return false;
    if (!g->exec && !record_sweep(s, a)) return false;
    g = find_sweep(s, a);
    if (!g || !g->exec) return false;
    if (hipGraphLaunch(g->exec, s->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphExecDestroy(g->exec);
        g->exec = nullptr;
        g->bad = true;
        return false;
    }
    g->stamp = ++s->graph_clock;
    s->last_error.clear();
    return true;
}

// After a successful call: remember the key (at most eight, least recently used out first).
static void note_sweep(XCSolver *s, const SweepArgs &a)
{
    if (SweepGraph *g = find_sweep(s, a)) {
        ++g->seen;
        g->stamp = ++s->graph_clock;
        return;
    }
    if (s->graphs.size() >= 8) {
        size_t old = 0;
        for (size_t i = 1; i < s->graphs.size(); ++i)
            if (s->graphs[i].stamp < s->graphs[old].stamp) old = i;
        if (s->graphs[old].exec) (void)hipGraphExecDestroy(s->graphs[old].exec);
        s->graphs.erase(s->graphs.begin() + (long)old);
    }
    SweepGraph g;
    g.ngrid = a.ngrid; g.nao = a.nao; g.nocc = a.nocc;
    g.dm = a.dm; g.ao = a.ao; g.grad = a.grad; g.w = a.w; g.vxc = a.vxc; g.cocc = a.cocc;
    g.seen = 1;
    g.stamp = ++s->graph_clock;
    s->graphs.push_back(g);
}

// The synchronous call: sweep + wait for Exc (shared by DFT_ComputeXC / DFT_ComputeXC64 / DFT_ComputeXCOcc)
static double xc_call_sync(XCSolver *s, long long ngrid, int nao, unsigned long long d_dm,
                           unsigned long long d_ao, unsigned long long d_ao_grad,
                           unsigned long long d_w, unsigned long long d_vxc,
                           unsigned long long d_cocc, int nocc)
{
    if (!s) return 0.0;
    DeviceGuard dg(s);
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (s->h_exc) *s->h_exc = nan; // before anything is enqueued: the last kernel overwrites it
    const SweepArgs a{(long)ngrid, nao, nocc, (const double *)d_dm, (const double *)d_ao, (const double *)d_ao_grad,
                      (const double *)d_w, (double *)d_vxc, (const double *)d_cocc};
    const bool graphed = !s->profile && s->h_exc_dev && ngrid > 0 && nao > 0 &&
                         (s->graph > 0 || (s->graph < 0 && (double)ngrid * nao <= GRAPH_AUTO_ELEMS));
    if (!(graphed && replay_sweep(s, a)) &&
        !xc_sweep(s, a.ngrid, nao, a.dm, a.ao, a.grad, a.w, a.vxc, true, a.cocc, nocc))
        return nan;
    if (graphed) note_sweep(s, a);
    if (s->h_exc_dev) { // Exc is written into host-mapped memory by the call's last kernel
        volatile double *hx = s->h_exc;
        if (s->spin_wait) {
            // The word is stored by the call's LAST kernel.  Default: that is the one-block k_finish_exc, which stream
            // order starts only after the Vxc reduce (and everything before it) has completed -- seeing the word means
            // the whole call is done, for consumers on any stream and for host reads, the reference's contract
            // (blocking 8-byte copy + cudaFree, dft_solver.cu:575-582).  With option "fuse_finish" = 1 the reduce
            // kernel's highest-index block stores it (no finishing launch): then it only orders consumers on the
            // SOLVER'S stream (the reference's pattern, d_vxc.get() on the null stream, dft.py:211), and
            // "strict_sync" = 1 adds a poll of the stream to completion for the others.  Both spins are bounded by
            // the stream state: a faulted kernel or an Exc that really is NaN ends them.
            for (unsigned spins = 1; std::isnan(*hx); ++spins) {
                if ((spins & 0xFFF) == 0 && hipStreamQuery(s->stream) != hipErrorNotReady) break;
                __builtin_ia32_pause();
            }
            if (s->strict_sync || std::isnan(*hx)) {
                hipError_t q;
                while ((q = hipStreamQuery(s->stream)) == hipErrorNotReady) __builtin_ia32_pause();
                if (!hip_ok(s, q, "XC sweep")) return nan;
            }
        } else if (!hip_ok(s, hipStreamSynchronize(s->stream), "synchronise")) {
            return nan;
        }
        const double out = *hx;
        if (std::isnan(out)) set_error(s, "Exc is NaN after the sweep completed (non-finite inputs, or the finishing kernel did not run)");
        return out;
    }
    double out = nan;
    if (!hip_ok(s, hipMemcpyAsync(&out, s->exc.p, sizeof(double), hipMemcpyDeviceToHost, s->stream), "copy Exc") ||
        !hip_ok(s, hipStreamSynchronize(s->stream), "synchronise"))
        return nan;
    if (std::isnan(out)) set_error(s, "Exc is NaN after the sweep completed (non-finite inputs)");
    return out;
}

double DFT_ComputeXC64(XCSolver *s, long long ngrid, int nao, unsigned long long d_dm,
                       unsigned long long d_ao, unsigned long long d_ao_grad,
                       unsigned long long d_w, unsigned long long d_vxc)
{
    return xc_call_sync(s, ngrid, nao, d_dm, d_ao, d_ao_grad, d_w, d_vxc, 0ULL, 0);
}

double DFT_ComputeXCOcc(XCSolver *s, long long ngrid, int nao, int nocc, unsigned long long d_cocc,
                        unsigned long long d_dm, unsigned long long d_ao, unsigned long long d_ao_grad,
                        unsigned long long d_w, unsigned long long d_vxc)
{
    if (s && !d_cocc) {
        set_error(s, "DFT_ComputeXCOcc needs the occupied orbitals");
        return std::numeric_limits<double>::quiet_NaN();
    }
    return xc_call_sync(s, ngrid, nao, d_dm, d_ao, d_ao_grad, d_w, d_vxc, d_cocc, nocc);
}

int DFT_ComputeXCOccAsync(XCSolver *s, long long ngrid, int nao, int nocc, unsigned long long d_cocc,
                          unsigned long long d_dm, unsigned long long d_ao, unsigned long long d_ao_grad,
                          unsigned long long d_w, unsigned long long d_vxc, unsigned long long d_exc)
{
    if (!s) return -1;
    DeviceGuard dg(s);
    if (!d_cocc) { set_error(s, "DFT_ComputeXCOccAsync needs the occupied orbitals"); return -1; }
    if (!xc_sweep(s, (long)ngrid, nao, (const double *)d_dm, (const double *)d_ao,
                  (const double *)d_ao_grad, (const double *)d_w, (double *)d_vxc, false, (const double *)d_cocc, nocc, (double *)d_exc))
        return -1;
    return 0;
}

double DFT_ComputeXC(XCSolver *s, int ngrid, int nao, unsigned long long d_dm,
                     unsigned long long d_ao, unsigned long long d_ao_grad,
                     unsigned long long d_w, unsigned long long d_vxc)
{
    return DFT_ComputeXC64(s, ngrid, nao, d_dm, d_ao, d_ao_grad, d_w, d_vxc);
}

int DFT_ComputeXCAsync(XCSolver *s, long long ngrid, int nao, unsigned long long d_dm,
                       unsigned long long d_ao, unsigned long long d_ao_grad,
                       unsigned long long d_w, unsigned long long d_vxc,
                       unsigned long long d_exc)
{
    if (!s) return -1;
    DeviceGuard dg(s);
    if (!xc_sweep(s, (long)ngrid, nao, (const double *)d_dm, (const double *)d_ao,
                  (const double *)d_ao_grad, (const double *)d_w, (double *)d_vxc, false, nullptr, 0, (double *)d_exc))
        return -1;
    return 0;
}

void DFT_ComputeCoulomb(XCSolver *s, int nao, unsigned long long d_eri, unsigned long long d_dm,
                        unsigned long long d_J)
{
    if (!s) return;
    DeviceGuard dg(s);
    jk(s, nao, (const double *)d_eri, (const double *)d_dm, (double *)d_J, nullptr);
}

void DFT_ComputeExchange(XCSolver *s, int nao, unsigned long long d_eri, unsigned long long d_dm,
                         unsigned long long d_K)
{
    if (!s) return;
    DeviceGuard dg(s);
    jk(s, nao, (const double *)d_eri, (const double *)d_dm, nullptr, (double *)d_K);
}

void DFT_ComputeJK(XCSolver *s, int nao, unsigned long long d_eri, unsigned long long d_dm,
                   unsigned long long d_J, unsigned long long d_K)
{
    if (!s) return;
    DeviceGuard dg(s);
    jk(s, nao, (const double *)d_eri, (const double *)d_dm, (double *)d_J, (double *)d_K);
}

int DFT_ComputeJKRows(XCSolver *s, int nao, int i_lo, int i_hi, unsigned long long d_eri_rows,
                      unsigned long long d_dm, unsigned long long d_J, unsigned long long d_K)
{
    if (!s) return -1;
    DeviceGuard dg(s);
    jk(s, nao, (const double *)d_eri_rows, (const double *)d_dm, (double *)d_J, (double *)d_K, i_lo, i_hi - i_lo);
    return s->last_error.empty() ? 0 : -1;
}

int DFT_ComputeJKFactorized(XCSolver *s, int nao, int naux, int nocc, unsigned long long d_chol,
                            unsigned long long d_dm, unsigned long long d_cocc, unsigned long long d_J,
                            unsigned long long d_K)
{
    if (!s) return -1;
    DeviceGuard dg(s);
    s->n_timed = 0;
    return jk_factorized(s, nao, naux, nocc, (const double *)d_chol, (const double *)d_dm,
                         (const double *)d_cocc, (double *)d_J, (double *)d_K);
}

int DFT_EvalAO(XCSolver *s, long long ngrid, int nao, int nshell, const double *shl_xyz,
               const int *shl_l, const int *shl_nprim, const int *shl_off, const int *shl_ao,
               const double *prim_exp, const double *prim_coef, int nprim_total,
               unsigned long long d_coords, unsigned long long d_ao, unsigned long long d_ao_grad)
{
    if (!s) return -1;
    DeviceGuard dg(s);
    s->last_error.clear();
    s->n_timed = 0;
    if (!s->device_ok) { set_error(s, "no usable HIP device"); return -1; }
    if (ngrid <= 0) { set_error(s, "bad AO sizes"); return -1; }
    AoTable tab;
    if (!prepare_ao_table(s, nao, nshell, shl_xyz, shl_l, shl_nprim, shl_off, shl_ao, prim_exp, prim_coef, nprim_total, tab)) return -1;
    ScopedTimer t(s, "eval_ao");
    launch_ao(s, tab, (long)ngrid, nao, (const double *)d_coords, (double *)d_ao, (double *)d_ao_grad);
    return hip_ok(s, hipGetLastError(), "AO launch") ? 0 : -1;
}

int DFT_ComputeXCDirect(XCSolver *s, long long ngrid, int nao, int nshell, const double *shl_xyz,
                        const int *shl_l, const int *shl_nprim, const int *shl_off, const int *shl_ao,
                        const double *prim_exp, const double *prim_coef, int nprim_total,
                        unsigned long long d_coords, unsigned long long d_w, unsigned long long d_dm,
                        unsigned long long d_vxc, unsigned long long d_exc, long long chunk_points)
{
    if (!s) return -1;
    DeviceGuard dg(s);
    s->last_error.clear();
    if (!s->device_ok) { set_error(s, "no usable HIP device"); return -1; }
    if (ngrid <= 0 || nao <= 0 || !d_coords || !d_w || !d_dm || !d_vxc) { set_error(s, "bad arguments to DFT_ComputeXCDirect"); return -1; }
    AoTable tab;
    if (!prepare_ao_table(s, nao, nshell, shl_xyz, shl_l, shl_nprim, shl_off, shl_ao, prim_exp, prim_coef, nprim_total, tab)) return -1;
    const bool gga = s->type != SOLVER_LDA;
    const int nplane = gga ? 4 : 1;
    // chunk: its planes stay within ~96 MB (they are written by the AO kernel and read twice right after: an
    // Infinity-Cache-sized working set), but never fewer than 64 sixteen-point tiles per CU; a multiple of 256
    long chunk = chunk_points > 0 ? (long)chunk_points : std::max<long>((long)(96.0e6 / (8.0 * nao * nplane)), 1024L * s->num_cu);
    chunk = std::min<long>(((chunk + 255) / 256) * 256, (long)ngrid);
    const size_t plane = (size_t)chunk * nao;
    if (!reserve(s, s->ao_ws, sizeof(double) * plane * nplane, "hipMalloc(AO chunk)") ||
        !reserve(s, s->vtmp, sizeof(double) * ((size_t)nao * nao + 1), "hipMalloc(V chunk)"))
        return -1;
    double *wao = (double *)s->ao_ws.p, *vt = (double *)s->vtmp.p, *eacc = vt + (size_t)nao * nao;
    const double *coords = (const double *)d_coords, *w = (const double *)d_w;
    const unsigned nb = (unsigned)(((size_t)nao * nao + 255) / 256);
    for (long g0 = 0; g0 < (long)ngrid; g0 += chunk) {
        const long n = std::min<long>(chunk, (long)ngrid - g0);
        double *wgr = gga ? wao + (size_t)n * nao : nullptr;       // (3, n, nao) right behind the values of THIS chunk
        launch_ao(s, tab, n, nao, coords + 3 * g0, wao, wgr);
        if (!xc_sweep(s, n, nao, (const double *)d_dm, wao, wgr, w + g0, vt, false)) return -1;
        hipLaunchKernelGGL(k_accumulate_chunk, dim3(nb), dim3(256), 0, s->stream, (long)nao * nao, g0 == 0 ? 0 : 1, vt,
                           (const double *)s->exc.p, (double *)d_vxc, eacc);
    }
    if (d_exc && !hip_ok(s, hipMemcpyAsync((void *)d_exc, eacc, sizeof(double), hipMemcpyDeviceToDevice, s->stream), "copy Exc"))
        return -1;
    return hip_ok(s, hipGetLastError(), "direct XC sweep launch") ? 0 : -1;
}

int DFT_SetOption(XCSolver *s, const char *key, double value)
{
    if (!s || !key) return -1;
    if (strcmp(key, "profile") && strcmp(key, "spin_wait") && strcmp(key, "strict_sync")) {
        DeviceGuard dg(s);
        drop_graphs(s); // recorded sweeps were launched under the old options
    }
    if (!strcmp(key, "eri_symmetric")) { s->eri_sym = value == 2.0 ? 2 : value != 0.0; return 0; }
    if (!strcmp(key, "graph")) { s->graph = value > 0.0 ? 1 : value < 0.0 ? -1 : 0; return 0; }
    if (!strcmp(key, "quirks")) { s->quirks = value != 0.0; return 0; }
    if (!strcmp(key, "path")) { s->path = (int)value; return 0; }
    if (!strcmp(key, "profile")) { s->profile = value != 0.0; return 0; }
    if (!strcmp(key, "spin_wait")) { s->spin_wait = value != 0.0; return 0; }
    if (!strcmp(key, "strict_sync")) { s->strict_sync = value != 0.0; return 0; }
    if (!strcmp(key, "fuse_finish")) { s->fuse_finish = value != 0.0; return 0; }
    if (!strcmp(key, "sweep_order")) { s->sweep_order = (int)value & 3; return 0; }
    if (!strcmp(key, "dbg")) { s->dbg = (int)value; return 0; }
    if (!strcmp(key, "occ")) { s->occ = value == 1.0 ? 1 : value == 2.0 ? 2 : 0; return 0; }
    if (!strcmp(key, "tiny")) { s->tiny = value > 0.0 ? 1 : value < 0.0 ? -1 : 0; return 0; }
    if (!strcmp(key, "ws_waves")) { s->ws_waves = value == 16.0 ? 16 : value == 8.0 ? 8 : 0; return 0; }
    if (!strcmp(key, "rho_rows")) { s->rho_rows = value == 128.0 ? 128 : 64; return 0; }
    if (!strcmp(key, "ao_pt")) { s->ao_pt = value == 16.0 ? 16 : value == 8.0 ? 8 : 0; return 0; }
    if (!strcmp(key, "ksplit")) { s->ksplit = value > 0 ? (int)value : 0; return 0; }
    return -1;
}

int DFT_SetStream(XCSolver *s, unsigned long long hip_stream)
{
    if (!s) return -1;
    DeviceGuard dg(s);
    if (s->device_ok) (void)hipStreamSynchronize(s->stream);
    s->stream = (hipStream_t)hip_stream;
    return 0; // recorded sweeps are stream-independent (recorded on the solver's own stream, launched on the current one)
}

const char *DFT_GetLastError(XCSolver *s)
{
    return s ? s->last_error.c_str() : "";
}

int DFT_GetTimings(XCSolver *s, double *ms, const char **names, int max_entries)
{
    if (!s || !s->device_ok) return 0;
    DeviceGuard dg(s);
    (void)hipStreamSynchronize(s->stream);
    int n = 0;
    for (size_t i = 0; i < s->n_timed && n < max_entries; ++i, ++n) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, s->timings[i].t0, s->timings[i].t1) != hipSuccess) t = -1.f;
        if (ms) ms[n] = t;
        if (names) names[n] = s->timings[i].name;
    }
    return n;
}

} // extern "C"
