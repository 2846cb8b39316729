// How much vector work fits NEXT to a saturating fp64-MFMA wave on one SIMD?
// Block = 256*(1+K) threads: waves 0-3 run back-to-back v_mfma_f64_16x16x4 (one per SIMD), waves 4.. are
// K "worker" waves per SIMD running one of several instruction mixes.  Reported per configuration:
// cycles per MFMA of the MFMA waves and cycles per instruction of ONE worker wave, plus the total
// worker instructions issued per MFMA per SIMD (the budget a fused kernel can spend per MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe2 tools/coissue_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: v_fma_f64, 8 independent chains      1: v_mad_u32, 8 chains
//      2: one dependent v_fma_f64 chain (latency-bound, like a transcendental's polynomial)
//      3: mix: 4 fma_f64 + 2 ds_read_b64 + 2 ds_write_b64 per iteration
template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, unsigned long long *clk, int mf_iters, int va_iters, int mfma_on, int prio)
{
    __shared__ double lds[8192];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-3;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = threadIdx.x * 1e-3, y = 1.0001;
        if (mfma_on)
            for (int i = 0; i < mf_iters; i++) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            }
        out[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        double r = 0;
        if (MODE == 0) {
            double v[8];
            for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
            for (int it = 0; it < va_iters; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_fma(v[i], 1.0000001, 1e-9);
            }
            for (int i = 0; i < 8; i++) r += v[i];
        }
        if (MODE == 1) {
            unsigned v[8];
            for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
            for (int it = 0; it < va_iters; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = v[i] * 3u + 7u;
            }
            unsigned s = 0;
            for (int i = 0; i < 8; i++) s += v[i];
            r = s;
        }
        if (MODE == 2) {
            double v = threadIdx.x;
            for (int it = 0; it < va_iters; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) v = __builtin_fma(v, 1.0000001, 1e-9);
            }
            r = v;
        }
        if (MODE == 3) {
            double v[4];
            for (int i = 0; i < 4; i++) v[i] = threadIdx.x + i;
            const int o = (threadIdx.x * 2) & 4095;
            for (int it = 0; it < va_iters; it++) {
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = __builtin_fma(v[i], 1.0000001, 1e-9);
                const double a = lds[o], b = lds[o + 4096];
                lds[o + 1] = v[0];
                lds[o + 4097] = v[1];
                v[2] += a;
                v[3] += b;
            }
            for (int i = 0; i < 4; i++) r += v[i];
        }
        out[blockIdx.x * 1024 + threadIdx.x] = r;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 16 + wave] = c1 - c0;
}

template <int MODE> void run(const char *name, int kw, int mfma_on, int prio, double *out, unsigned long long *clk)
{
    const int mf = 2000, va = 2000;
    const int per_iter = MODE == 3 ? 10 : 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256 * (1 + kw)), 0, 0, out, clk, mf, va, mfma_on, prio);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; b++)
        for (int w = 0; w < 4 * (1 + kw); w++) {
            if (w < 4) m += h[b * 16 + w];
            else v += h[b * 16 + w];
        }
    m /= 1024;
    v /= 1024.0 * kw;
    const double cyc_mfma = m / (4.0 * mf), cyc_instr = v / ((double)per_iter * va);
    // worker instructions per MFMA per SIMD while both run: kw workers * (cycles per MFMA / cycles per instr)
    printf("%-10s workers/SIMD=%d mfma=%d prio=%d : %.1f cyc/MFMA ; worker %.1f cyc/instr ; %.2f worker instr per MFMA per SIMD\n",
           name, kw, mfma_on, prio, cyc_mfma, cyc_instr, mfma_on ? kw * cyc_mfma / cyc_instr : 0.0);
}

int main()
{
    double *out;
    hipMalloc(&out, 256 * 1024 * 8);
    unsigned long long *clk;
    hipMalloc(&clk, 256 * 16 * 8);
    for (int kw = 1; kw <= 3; ++kw) {
        run<0>("fma_f64x8", kw, 0, 0, out, clk);
        run<0>("fma_f64x8", kw, 1, 0, out, clk);
        run<0>("fma_f64x8", kw, 1, 1, out, clk);
        run<1>("mad_u32x8", kw, 1, 0, out, clk);
        run<2>("fma_chain", kw, 0, 0, out, clk);
        run<2>("fma_chain", kw, 1, 0, out, clk);
        run<3>("mix_lds", kw, 0, 0, out, clk);
        run<3>("mix_lds", kw, 1, 0, out, clk);
    }
    return 0;
}
