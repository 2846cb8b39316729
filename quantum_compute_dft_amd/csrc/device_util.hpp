// Small device helpers shared by the kernel headers.
#pragma once
#include <hip/hip_runtime.h>

namespace qcdft {

// LDS-only workgroup barrier, spelled out: exactly what hipcc emits for __syncthreads() on gfx950
// (s_waitcnt lgkmcnt(0); s_barrier -- checked in the ISA; no vmcnt wait outside tgsplit mode), written
// as asm where a kernel's correctness argument depends on global accesses staying in flight across it.
// The coupling to watch is elsewhere: gfx9 counts loads and stores on the one vmcnt and retires them in
// order, so a load issued after stores cannot be waited for without waiting for those stores too, and
// a wait the compiler cannot count (conditional accesses, unknown trip counts) becomes vmcnt(0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

} // namespace qcdft
