// Issue rate of SCALAR instructions (and scalar/vector mixes) of a worker wave next to a saturating
// fp64-MFMA wave on the same SIMD.  Inline asm so the instruction counts are exact.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/coissue_probe3 tools/coissue_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE> // 0: 16 s_add_u32   1: 16 v_add_u32   2: 8 s_add + 8 v_add alternating   3: 16 s_mul_i32
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *clk, int mf_iters, int va_iters, int mfma_on)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4, v0 = threadIdx.x, v1 = 5, v2 = 6, v3 = 7;
        for (int it = 0; it < va_iters; it++) {
            if (MODE == 0)
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
            if (MODE == 1)
                asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                             "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                             "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                             "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (MODE == 2)
                asm volatile("s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %5, %5, 1\n"
                             "s_add_u32 %2, %2, 1\n v_add_u32 %6, %6, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %7, %7, 1\n"
                             "s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %5, %5, 1\n"
                             "s_add_u32 %2, %2, 1\n v_add_u32 %6, %6, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %7, %7, 1\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"scc");
            if (MODE == 3)
                asm volatile("s_mul_i32 %0, %0, 3\n s_mul_i32 %1, %1, 3\n s_mul_i32 %2, %2, 3\n s_mul_i32 %3, %3, 3\n"
                             "s_mul_i32 %0, %0, 3\n s_mul_i32 %1, %1, 3\n s_mul_i32 %2, %2, 3\n s_mul_i32 %3, %3, 3\n"
                             "s_mul_i32 %0, %0, 3\n s_mul_i32 %1, %1, 3\n s_mul_i32 %2, %2, 3\n s_mul_i32 %3, %3, 3\n"
                             "s_mul_i32 %0, %0, 3\n s_mul_i32 %1, %1, 3\n s_mul_i32 %2, %2, 3\n s_mul_i32 %3, %3, 3\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
        }
        out[blockIdx.x * 512 + threadIdx.x] = s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = c1 - c0;
}
template <int MODE> void run(const char *name, int mfma_on, double *out, unsigned long long *clk)
{
    const int mf = 2000, va = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, clk, mf, va, mfma_on); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; b++) for (int w = 0; w < 8; w++) { if (w < 4) m += h[b * 8 + w]; else v += h[b * 8 + w]; }
    m /= 1024; v /= 1024;
    printf("%-16s mfma=%d : %.1f cyc/MFMA ; worker %.2f cyc/instr (16 per iteration + loop control)\n", name, mfma_on, m / (4.0 * mf), v / (16.0 * va));
}
int main()
{
    double *out; hipMalloc(&out, 256 * 512 * 8);
    unsigned long long *clk; hipMalloc(&clk, 256 * 8 * 8);
    run<0>("s_add_u32", 0, out, clk); run<0>("s_add_u32", 1, out, clk);
    run<1>("v_add_u32", 0, out, clk); run<1>("v_add_u32", 1, out, clk);
    run<2>("s_add/v_add mix", 0, out, clk); run<2>("s_add/v_add mix", 1, out, clk);
    run<3>("s_mul_i32", 0, out, clk); run<3>("s_mul_i32", 1, out, clk);
    return 0;
}
