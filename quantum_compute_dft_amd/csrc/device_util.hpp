// Small device helpers shared by the kernel headers.
#pragma once
#include <hip/hip_runtime.h>

namespace qcdft {

// LDS-only workgroup barrier.  __syncthreads() also drains every outstanding global access
// (s_waitcnt vmcnt(0)): stores in flight and loads prefetched for later stages would all be
// waited for at every barrier.  gfx9 counts loads and stores on the one vmcnt and retires them in
// order, so code around this barrier has to keep its waits counted (a fixed number of younger
// accesses), never vmcnt(0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

} // namespace qcdft
