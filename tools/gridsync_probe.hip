// Cost of a grid-wide barrier between dependent phases on gfx950: NWG workgroups (one per CU at most), each phase writes a
// value per workgroup to memory, the barrier, every workgroup reads all of them (agent-scope fences either side: the
// workgroups sit on different XCDs with different L2s).  build: hipcc --offload-arch=gfx950 -O3 tools/gridsync_probe.hip -o tools/gridsync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool gsync(unsigned *bar, unsigned nwg, int *err)
{
    __threadfence();
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) { *err = 1; ok = false; break; }   // ~seconds: a lost workgroup must not hang the GPU
            }
        }
    }
    __syncthreads();
    __threadfence();
    return ok;
}

__global__ void k_probe(unsigned *bar, double *buf, int phases, int *err, double *out)
{
    const unsigned nwg = gridDim.x;
    double acc = 0.0;
    for (int p = 0; p < phases; ++p) {
        if (threadIdx.x == 0) buf[(p & 1) * 1024 + blockIdx.x] = (double)(p + blockIdx.x);
        if (!gsync(bar, nwg, err)) return;
        double s = 0.0;
        for (unsigned w = threadIdx.x; w < nwg; w += blockDim.x) s += buf[(p & 1) * 1024 + w];
        acc += s;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main()
{
    unsigned *bar; double *buf, *out; int *err;
    hipMalloc(&bar, 64); hipMemset(bar, 0, 64);
    hipMalloc(&buf, 2048 * 8); hipMalloc(&out, 1024 * 8); hipMalloc(&err, 4); hipMemset(err, 0, 4);
    const int phases = 200;
    for (int nwg : {1, 8, 16, 32, 64, 128, 256}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(bar, 0, 64);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            void *args[] = {&bar, &buf, (void *)&phases, &err, &out};
            hipEventRecord(e0, 0);
            hipError_t rc = hipLaunchCooperativeKernel((const void *)k_probe, dim3(nwg), dim3(512), args, 0, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            std::vector<double> h(nwg); hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost);
            double expect = 0; for (int p = 0; p < phases; ++p) for (int w = 0; w < nwg; ++w) expect += p + w;
            if (rep) printf("nwg %3d: rc %d err %d  %.2f us per phase (barrier + read-all)  sum ok %d\n", nwg, (int)rc, herr, 1e3 * ms / phases, h[0] == expect && h[nwg - 1] == expect);
        }
    }
    return 0;
}
