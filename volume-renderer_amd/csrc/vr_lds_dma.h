// vr_lds_dma.h -- gfx950 helpers shared by the LDS-staged ray-march kernel (vr_tslab.hip):
// LDS-DMA (global_load_lds_dwordx4 through M0), explicit vmcnt waits, wave-uniform values, DPP reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vr {

#define VR_LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ uint32_t lds_offset_of(const void *p) { return (uint32_t)(size_t)(VR_LDS_AS const char *)p; }

// one 1-KiB piece of LDS-DMA: lane l's 16 bytes from gsrc land at lds_dst + 16*l (lds_dst is
// wave-uniform and goes through M0; lanes switched off by EXEC leave their 16 bytes alone).
// The compiler does not count this load: completion is waited for with slab_wait_pieces().
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// the same with the source as a wave-uniform 64-bit base (SGPR pair) + a per-lane 32-bit byte offset: the address
// arithmetic of a piece is then 32-bit per lane, the 64-bit part is scalar
__device__ __forceinline__ void glds16_rel(const void *gbase, uint32_t rel, uint32_t lds_dst)
{
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(rel), "s"(gbase), "s"(lds_dst) : "memory");
}

// wait until at most n of this wavefront's vector-memory operations are outstanding (they
// complete in order, so everything issued before the last n has landed)
__device__ __forceinline__ void slab_wait_pieces(int n)
{
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

__device__ __forceinline__ float uniform_f(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ int uniform_i(int v) { return (int)__builtin_amdgcn_readfirstlane((uint32_t)v); }
// wavefront-wide min / max of a float (result uniform): four DPP steps inside each row of 16 lanes, then
// the four rows through readlane -- 11 instructions, no LDS crossbar
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l)); }
__device__ __forceinline__ float wave_min_f(float v)
{
    v = fminf(v, dpp_f<0xB1>(v));      // quad_perm [1,0,3,2]
    v = fminf(v, dpp_f<0x4E>(v));      // quad_perm [2,3,0,1]
    v = fminf(v, dpp_f<0x141>(v));     // row_half_mirror
    v = fminf(v, dpp_f<0x140>(v));     // row_mirror
    return fminf(fminf(readlane_f(v, 0), readlane_f(v, 16)), fminf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ __forceinline__ float wave_max_f(float v)
{
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// integer wave reductions (every lane of the wavefront must execute them: inactive lanes would leave stale registers behind
// the read-lanes).  Same-address LDS atomics cost ~18 cycles per LANE on gfx950 (measured: 7 atomics by 256 threads = 3 %
// of a 480-us tile), so reductions go through these first and one lane per wavefront does the atomic.
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_min_i(int v)
{
    v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v)
{
    v = max(v, dpp_i<0xB1>(v)); v = max(v, dpp_i<0x4E>(v)); v = max(v, dpp_i<0x141>(v)); v = max(v, dpp_i<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
template <typename T> __device__ __forceinline__ T sel3(int ax, T v0, T v1, T v2) { return ax == 0 ? v0 : (ax == 1 ? v1 : v2); }

}  // namespace vr
