// the headline instance of raymarch_relay_kernel alone, for ISA inspection:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S -o /tmp/relay.s tools/isa/probe_relay.hip
#define VR_TU 99
#include "../../volume-renderer_amd/csrc/vr_kernels.hip"
namespace vr {
template __global__ void raymarch_relay_kernel<uint16_t, 1, 0, true, true, true, true, true, 0, 0>(
    const FrameParams, const uint16_t *, const float4 *, const uint32_t, float4 *, uint32_t *, const uint32_t *, const void *, const uint32_t);
}
