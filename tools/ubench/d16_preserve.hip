// does a d16 LDS load keep the other half of its destination on this chip?  (gfx950 code objects are built for sramecc "any":
// the compiler never selects the tied d16 forms; with SRAM ECC on the hardware zeroes the unused half)
//   hipcc --offload-arch=gfx950 -O3 d16_preserve.hip -o d16_preserve && ./d16_preserve
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    __shared__ unsigned short ring[64];
    ring[threadIdx.x] = (unsigned short)(1000 + threadIdx.x);
    __syncthreads();
    unsigned r = 0x4B000000u, addr = threadIdx.x * 2;
    asm volatile("ds_read_u16_d16 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(addr));
    out[threadIdx.x] = r;
}
int main()
{
    unsigned *d, h[64];
    (void)hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_read_u16_d16 into a register holding 0x4B000000: lane 0 -> 0x%08x, lane 5 -> 0x%08x (%s)\n", h[0], h[5], (h[5] >> 16) == 0x4B00 ? "high half kept" : "high half NOT kept");
    return 0;
}
