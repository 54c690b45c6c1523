// the config-4 grey instance (u8, bricked, 64-bit offsets, address tables) alone, for ISA inspection:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S -o /tmp/cfg4.s tools/isa/probe_cfg4.hip
#define VR_TU 99
#include "../../volume-renderer_amd/csrc/vr_kernels.hip"
namespace vr {
template __global__ void raymarch_fast_kernel<uint8_t, 1, 0, 0, true, true, true, true, 0, false, 8, true, false>(
    const FrameParams, const uint8_t *, const float4 *, const uint32_t, float4 *, uint32_t *, const unsigned, const unsigned,
    const unsigned, const uint32_t *, const uint16_t *, const uint32_t, const void *, const uint32_t);
}
