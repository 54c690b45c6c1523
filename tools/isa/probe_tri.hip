// the cfg3 instance of raymarch_tri_kernel (u16, bricked, POW2, composite, apron copy) alone, for ISA inspection
#define VR_TU 99
#include "../../volume-renderer_amd/csrc/vr_kernels.hip"
namespace vr {
template __global__ void raymarch_tri_kernel<uint16_t, 1, 0, 0, true, 0, true>(const FrameParams, const uint16_t *, const uint32_t, float4 *, uint32_t *,
                                                                              const uint32_t *);
}
