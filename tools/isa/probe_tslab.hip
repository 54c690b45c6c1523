// the cfg3 instance of the LDS-staged TRILINEAR kernel (u16, POW2, composite) alone, for ISA inspection:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -S -o /tmp/tslab.s tools/isa/probe_tslab.hip
#define VR_TSLAB_TU 99
#include "../../volume-renderer_amd/csrc/vr_tslab.hip"
namespace vr {
#ifndef PROBE_NW
#define PROBE_NW 8
#define PROBE_LDSKB 80
#endif
#ifndef PROBE_PERM
#define PROBE_PERM false
#endif
template __global__ void raymarch_tslab_kernel<uint16_t, 0, 0, true, 0, PROBE_NW, PROBE_LDSKB, PROBE_PERM>(const FrameParams, const uint16_t *, const uint8_t *, const uint8_t *, const uint8_t *, const float4 *, float4 *, uint32_t *,
                                                                                               const uint32_t *, const int, const uint16_t *);
}
