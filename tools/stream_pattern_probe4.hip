// Probe D of stream_pattern_probe3 (workgroup = 16 rows, wave w = 32-column block w; 85 us) grown towards
// k_rho_occ_rs one feature at a time, to find what costs the kernel its last 15 us:
//   PF   cross-tile prefetch (next tile's AO group issued when this one's is consumed, same for the gradients)
//   LDS  AO group staged to wave-private LDS and read back as eight fragments
//   BAR  two workgroup barriers per tile + a 4-way exchange of two 32-byte slots per lane through LDS
//   ST   per-row results stored (5 doubles per grid row)
// Build: hipcc --offload-arch=gfx950 -O3 -w -o tools/stream_pattern_probe4 tools/stream_pattern_probe4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_double2(__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2]));
}
template <bool PF, bool LDS, bool BAR, bool ST>
__global__ __launch_bounds__(256) void k_stream(long ngrid, int nao, const double *__restrict__ p0, const double *__restrict__ p1,
                                                const double *__restrict__ p2, const double *__restrict__ p3, double *out)
{
    __shared__ __attribute__((aligned(32))) double sh[4 * 528 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, q = lane >> 4;
    const long plane = ngrid * (long)nao, ntile = (ngrid + 15) / 16;
    unsigned voff[4];
    for (int r = 0; r < 4; ++r) voff[r] = (unsigned)((4 * r + q) * nao + 2 * li) * 8u;
    double *As = sh + wave * 528;
    double s = 0;
    auto rs = [&](const double *p, long tile) {
        const bool in = tile < ntile;
        const long e0 = (in ? tile : 0) * 16 * (long)nao;
        const long rem = (plane - e0) * 8;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p + e0), 0, in ? (unsigned)(rem > 0xFFFFFFFFL ? 0xFFFFFFFFL : rem) : 0u, 0x00020000);
    };
    double2 a[4], g[3][4];
    auto ia = [&](long tile) {
        const __amdgpu_buffer_rsrc_t r0 = rs(p0, tile);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = ld(r0, voff[r], (unsigned)(wave * 256));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto ig = [&](long tile) {
        const __amdgpu_buffer_rsrc_t r1 = rs(p1, tile), r2 = rs(p2, tile), r3 = rs(p3, tile);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            g[0][r] = ld(r1, voff[r], (unsigned)(wave * 256));
            g[1][r] = ld(r2, voff[r], (unsigned)(wave * 256));
            g[2][r] = ld(r3, voff[r], (unsigned)(wave * 256));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (PF) { ia(blockIdx.x); ig(blockIdx.x); }
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        if (!PF) ia(tile);
        double t = 0;
        if (LDS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { As[(4 * r + q) * 33 + 2 * li] = a[r].x; As[(4 * r + q) * 33 + 2 * li + 1] = a[r].y; }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 8; ++k) t += As[li * 33 + 8 * q + k];
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) t += a[r].x + a[r].y;
        }
        if (PF) ia(tile + gridDim.x);
        if (!PF) ig(tile);
        if (BAR) {
            *reinterpret_cast<d4 *>(As + lane * 4) = (d4){t, t, t, t};
            *reinterpret_cast<d4 *>(As + (64 + lane) * 4) = (d4){t, t, t, t};
            __syncthreads();
            d4 u = (d4){0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < 4; ++w) u += *reinterpret_cast<const d4 *>(sh + w * 528 + lane * 4) + *reinterpret_cast<const d4 *>(sh + w * 528 + (64 + lane) * 4);
            __syncthreads();
            t = u[0] + u[1] + u[2] + u[3];
        }
        if (ST && wave == 1 && lane < 16 && tile * 16 + lane < ngrid) out[tile * 16 + lane] = t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t += g[0][r].x + g[1][r].y + g[2][r].x;
        if (PF) ig(tile + gridDim.x);
        if (ST && wave == 0 && lane < 16 && tile * 16 + lane < ngrid) {
            out[ngrid + 4 * (tile * 16 + lane) + 0] = t; out[ngrid + 4 * (tile * 16 + lane) + 1] = t;
            out[ngrid + 4 * (tile * 16 + lane) + 2] = t; out[ngrid + 4 * (tile * 16 + lane) + 3] = t;
        }
        s += t;
    }
    if (s == 1.234e-300) out[0] = s;
}
template <bool PF, bool LDS, bool BAR, bool ST> void run(const char *what, int wgs_per_cu, int ncu, long ngrid, int nao, double *p, double *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    size_t plane = (size_t)ngrid * nao;
    int blocks = ncu * wgs_per_cu;
    auto L = [&] { hipLaunchKernelGGL((k_stream<PF, LDS, BAR, ST>), dim3(blocks), dim3(256), 0, 0, ngrid, nao, p, p + plane, p + 2 * plane, p + 3 * plane, out); };
    for (int r = 0; r < 200; r++) L();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 50; r++) L();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
    printf("%-40s WGs/CU=%d: %.1f us  %.0f GB/s\n", what, wgs_per_cu, ms * 1e3, 4.0 * plane * 8 / ms * 1e-6);
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); int ncu = prop.multiProcessorCount;
    long ngrid = 143556; int nao = 114; size_t plane = (size_t)ngrid * nao;
    double *p; hipMalloc(&p, plane * 8 * 4 + 4096); double *out; hipMalloc(&out, 8 * 5 * ngrid + 64);
    std::vector<double> h(plane * 4);
    srand(1); for (auto &x : h) x = (rand() / (double)RAND_MAX - 0.5) * 0.8; hipMemcpy(p, h.data(), plane * 8 * 4, hipMemcpyHostToDevice);
    for (int w : {2, 3}) {
        run<false, false, false, false>("D  plain", w, ncu, ngrid, nao, p, out);
        run<true, false, false, false>("D + prefetch", w, ncu, ngrid, nao, p, out);
        run<false, true, false, false>("D + LDS staging", w, ncu, ngrid, nao, p, out);
        run<false, false, true, false>("D + barriers/exchange", w, ncu, ngrid, nao, p, out);
        run<false, false, false, true>("D + stores", w, ncu, ngrid, nao, p, out);
        run<true, true, true, true>("D + all four", w, ncu, ngrid, nao, p, out);
        run<false, true, true, true>("D + all but prefetch", w, ncu, ngrid, nao, p, out);
    }
    return 0;
}
