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

// fp64 MFMA operand maps (cdna_hip_programming.md section 3): lane l holds A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15]; result register r holds D[row = (l>>4) + 4r][col = l&15].
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Plane tiles are read with BUFFER loads: a wave-uniform descriptor (SGPRs), a wave-uniform byte
// offset of the tile (one SGPR), one 32-bit per-thread byte offset and immediate column offsets.
// The hardware range check returns zeros for rows past the grid, so the loader stream carries no
// masks, clamps or 64-bit address arithmetic -- it has to fit in the few issue slots a wave gets
// next to a saturating MFMA wave on the same SIMD (measured, tools/coissue_probe*.hip).
// Columns >= nao of a staged row hold finite data of the next row; they only ever multiply
// exact zeros (zero-padded Ds rows / discarded V tiles).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ONE descriptor per plane for the whole kernel (base = plane start, range = the plane); the sub-tile is
// selected by the SGPR offset of the load, which the hardware adds to the address AND to the range check
// (tools/bufrange_probe.hip: lanes with voffset + soffset >= num_records read 0 on gfx950).  A drain step
// passes soffset = num_records: every lane is out of range, no memory traffic, the load still counts in
// vmcnt, so the loader loop stays branch-free.  This replaced a per-step, per-plane descriptor rebuild
// (~110 scalar instructions per sub-tile): next to a saturating fp64-MFMA wave a wave issues ONE scalar
// instruction per 16 cycles and one vector instruction per ~24 (tools/coissue_probe3.hip,
// profiles/r02_coissue_probe3.txt), so those scalar instructions alone cost ~1800 of the 4096 cycles the
// matrix pipe needs per sub-tile and the loaders -- not HBM, not the MFMAs -- set the pace.
// Planes of 4 GiB or more do not fit a descriptor range: the host routes them to the generic kernels.
// Per-tile descriptor (base = the tile's first element, range = to the end of the plane; `live` = false
// gives zero records): the form the large-basis and Cholesky kernels use, whose planes may exceed 4 GiB and
// whose MFMA loops are long enough to hide the rebuild.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_tile_rsrc(const double *plane, long plane_elems,
                                                                  long first_elem, bool live = true)
{
    const long remain = (plane_elems - first_elem) * 8; // bytes to the end of the plane (> 0)
    const unsigned nrec = !live ? 0u : remain > 0xFFFFFFFFL ? 0xFFFFFFFFu : (unsigned)remain;
    return __builtin_amdgcn_make_buffer_rsrc((void *)(plane + first_elem), 0, nrec, 0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const double *plane, long plane_elems)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)plane, 0, (unsigned)(plane_elems * 8), 0x00020000);
}
// pair (c, c+1) of one row: `voff` = byte offset of (row, 2*seg) in the tile, IMM = byte offset of the
// column group, `soff` = byte offset of the tile in the plane
template <bool VEC, int IMM>
__device__ __forceinline__ void buf_load_pair(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double &a, double &b)
{
    if (VEC) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + IMM, soff, 0);
        a = __hiloint2double((int)v[1], (int)v[0]);
        b = __hiloint2double((int)v[3], (int)v[2]);
    } else { // odd nao or 8-byte aligned base: two 8-byte loads
        a = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff + IMM, soff, 0));
        b = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff + IMM + 8, soff, 0));
    }
}
// JN column groups of GW bytes each (GW = 256: 16 lanes per grid row, 512: 32 lanes per row)
template <int JN, bool VEC, int GW = 256, int J = 0>
__device__ __forceinline__ void buf_load_row(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double (&dst)[2 * JN])
{
    if constexpr (J < JN) {
        buf_load_pair<VEC, GW * J>(r, voff, soff, dst[2 * J], dst[2 * J + 1]);
        buf_load_row<JN, VEC, GW, J + 1>(r, voff, soff, dst);
    }
}
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
// Sum over the 16 lanes of a DPP row with row rotations: pure VALU, no LDS traffic (the
// ds_bpermute butterfly cost 21 us of a 147 us kernel).  Every lane ends with the total.
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_mov_f64<0x128>(v); // row_ror:8
    v += dpp_mov_f64<0x124>(v); // row_ror:4
    v += dpp_mov_f64<0x122>(v); // row_ror:2
    v += dpp_mov_f64<0x121>(v); // row_ror:1
    return v;
}

__device__ __forceinline__ double2 buf_load_d2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_double2(__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2]));
}
__device__ __forceinline__ double buf_load_d1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
// two consecutive doubles at byte offset voff+soff; VEC = 16-byte aligned rows (nao even)
template <bool VEC>
__device__ __forceinline__ double2 buf_load_pair2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    if (VEC) return buf_load_d2(r, voff, soff);
    return make_double2(buf_load_d1(r, voff, soff), buf_load_d1(r, voff + 8, soff));
}

} // namespace qcdft
