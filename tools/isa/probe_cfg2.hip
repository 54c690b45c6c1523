// compiles ONLY the cfg2-shape instance of raymarch_fast_kernel (512x512x452 u16: certified divisions instead of power-of-two scales,
// window clamp in the classification, pipelined loop, packed copy) for ISA inspection:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S -o /tmp/cfg2.s tools/isa/probe_cfg2.hip
#define VR_TU 99
#include "../../volume-renderer_amd/csrc/vr_kernels.hip"
namespace vr {
template __global__ void raymarch_fast_kernel<uint16_t, 1, 1, 0, false, true, false, false, 0, false, 8, true, true, true>(
    const FrameParams, const uint16_t *, const float4 *, const uint32_t, float4 *, uint32_t *, const unsigned, const unsigned,
    const unsigned, const uint32_t *, const uint16_t *, const uint32_t, const void *, const uint32_t);
}
