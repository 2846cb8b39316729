// Is the SGPR offset of a raw buffer load part of the hardware range check on gfx950?
// Buffer of N doubles (value 1+i) followed by sentinels (-7); descriptor covers exactly the N doubles.
// Loads at voffset = 8*lane with soffset = 8*(N-32): lanes 0-31 in range, lanes 32-63 past the end.
// Expected if soffset takes part in the check: lanes >= 32 read 0.  Otherwise they read the sentinel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const double *p, unsigned nrec, unsigned soff, double *out)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, nrec, 0x00020000);
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, threadIdx.x * 8u, soff, 0);
    out[threadIdx.x] = __hiloint2double((int)v[1], (int)v[0]);
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, threadIdx.x * 8u, nrec, 0);   // soffset == num_records: all out
    out[64 + threadIdx.x] = __hiloint2double((int)w[1], (int)w[0]);
    const u32x2 z = __builtin_amdgcn_raw_buffer_load_b64(r, threadIdx.x * 8u, 0xFFFFFFF0u, 0); // wrap-around attempt
    out[128 + threadIdx.x] = __hiloint2double((int)z[1], (int)z[0]);
}
int main()
{
    const int N = 1024;
    std::vector<double> h(N + 256, -7.0);
    for (int i = 0; i < N; ++i) h[i] = 1.0 + i;
    double *p, *out;
    hipMalloc(&p, h.size() * 8); hipMalloc(&out, 192 * 8);
    hipMemcpy(p, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, (unsigned)(N * 8), (unsigned)(8 * (N - 32)), out);
    std::vector<double> o(192);
    hipMemcpy(o.data(), out, 192 * 8, hipMemcpyDeviceToHost);
    printf("soffset=8*(N-32): lane 31 -> %g (expect %d), lane 32 -> %g, lane 63 -> %g  (0 = soffset is range-checked, -7 = it is not)\n", o[31], N, o[32], o[63]);
    printf("soffset=num_records: lane 0 -> %g, lane 63 -> %g (expect 0)\n", o[64], o[127]);
    printf("soffset=0xFFFFFFF0: lane 0 -> %g, lane 2 -> %g, lane 63 -> %g\n", o[128], o[130], o[191]);
    return 0;
}
