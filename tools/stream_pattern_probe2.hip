// Does the attainable HBM stream rate of the plane pattern depend on (a) the DATA (zero-filled vs random
// fp64: the r01 probe streamed a hipMemset buffer), (b) lanes per row (16 = 256 B runs, 32 = 512 B runs),
// (c) the workgroup size?  Four planes of Benzene/def2-SVP (nao 114, 143 556 rows), persistent
// workgroups, tile t of workgroup b = b + k*grid, depth-1 register prefetch as in stream_pattern_probe.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/stream_pattern_probe2 tools/stream_pattern_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int THREADS, int SEGS> // SEGS lanes per grid row, each lane 16 B per 2*SEGS-column group
__global__ __launch_bounds__(THREADS) void k_stream(long ngrid, int nao, const double *__restrict__ p0, const double *__restrict__ p1,
                                                    const double *__restrict__ p2, const double *__restrict__ p3, double *out)
{
    constexpr int ROWS = THREADS / SEGS, NJ = 128 / (2 * SEGS);
    const int row = threadIdx.x / SEGS, seg = threadIdx.x % SEGS;
    const long ntile = (ngrid + ROWS - 1) / ROWS;
    double s = 0;
    double2 v[4 * NJ];
    long t = blockIdx.x;
    auto issue = [&](long tt) {
        long g = tt * ROWS + row;
        bool ok = tt < ntile && g < ngrid;
        size_t ro = (size_t)(ok ? g : 0) * nao;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            int c = 2 * SEGS * j + 2 * seg;
            int cc = c < nao - 2 ? c : nao - 2;
            v[4 * j + 0] = *(const double2 *)(p0 + ro + cc);
            v[4 * j + 1] = *(const double2 *)(p1 + ro + cc);
            v[4 * j + 2] = *(const double2 *)(p2 + ro + cc);
            v[4 * j + 3] = *(const double2 *)(p3 + ro + cc);
        }
    };
    issue(t);
    for (; t < ntile; t += gridDim.x) {
#pragma unroll
        for (int i = 0; i < 4 * NJ; i++) s += v[i].x + v[i].y;
        issue(t + gridDim.x);
    }
    if (s == 1.234e-300) out[0] = s;
}
template <int THREADS, int SEGS> void run(const char *what, int wgs_per_cu, int ncu, long ngrid, int nao, double *p, double *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    size_t plane = (size_t)ngrid * nao;
    int blocks = ncu * wgs_per_cu;
    auto L = [&] { hipLaunchKernelGGL((k_stream<THREADS, SEGS>), dim3(blocks), dim3(THREADS), 0, 0, ngrid, nao, p, p + plane, p + 2 * plane, p + 3 * plane, out); };
    for (int r = 0; r < 200; r++) L(); // clock spin-up
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 50; r++) L();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
    printf("%-7s threads=%4d lanes/row=%2d WGs/CU=%d: %.1f us  %.0f GB/s\n", what, THREADS, SEGS, wgs_per_cu, ms * 1e3, 4.0 * plane * 8 / ms * 1e-6);
}
int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); int ncu = prop.multiProcessorCount;
    long ngrid = 143556; int nao = 114; size_t plane = (size_t)ngrid * nao;
    double *p; hipMalloc(&p, plane * 8 * 4 + 4096); double *out; hipMalloc(&out, 64);
    std::vector<double> h(plane * 4);
    for (int mode = 0; mode < 2; ++mode) {
        const char *what = mode ? "random" : "zeros";
        if (mode) { srand(1); for (auto &x : h) x = (rand() / (double)RAND_MAX - 0.5) * 0.8; hipMemcpy(p, h.data(), plane * 8 * 4, hipMemcpyHostToDevice); }
        else hipMemset(p, 0, plane * 8 * 4);
        run<512, 16>(what, 1, ncu, ngrid, nao, p, out);
        run<512, 32>(what, 1, ncu, ngrid, nao, p, out);
        run<256, 16>(what, 1, ncu, ngrid, nao, p, out);
        run<256, 16>(what, 2, ncu, ngrid, nao, p, out);
        run<1024, 16>(what, 1, ncu, ngrid, nao, p, out);
        run<1024, 32>(what, 1, ncu, ngrid, nao, p, out);
        run<256, 16>(what, 4, ncu, ngrid, nao, p, out);
    }
    return 0;
}
