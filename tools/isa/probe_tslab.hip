// the cfg3 instance of the LDS-staged TRILINEAR kernel (u16, POW2, composite) alone, for ISA inspection
#define VR_SLAB_TU 99
#include "../../volume-renderer_amd/csrc/vr_slab.hip"
namespace vr {
template __global__ void raymarch_slab_kernel<uint16_t, false, 0, 0, true, false, 0, true>(const FrameParams, const uint16_t *, const uint8_t *, const float4 *,
                                                                                            float4 *, uint32_t *, const uint32_t *);
}
