// compiles ONLY the headline instance of raymarch_fast_kernel (for ISA inspection; seconds instead of minutes):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S -o /tmp/headline.s tools/isa/probe_headline.hip
#define VR_TU 99
#include "../../volume-renderer_amd/csrc/vr_kernels.hip"
namespace vr {
template __global__ void raymarch_fast_kernel<uint16_t, 1, 0, 0, false, true, true, true, 0, false, 8, true, true>(
    const FrameParams, const uint16_t *, const float4 *, const uint32_t, float4 *, uint32_t *, const unsigned, const unsigned,
    const unsigned, const uint32_t *, const uint16_t *, const uint32_t, const void *, const uint32_t);
}
