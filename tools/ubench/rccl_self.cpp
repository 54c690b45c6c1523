// probe: does a grouped ncclSend + ncclRecv to SELF move the bytes on this box, with both calls on one stream and
// with the send and the receive on different streams?   hipcc rccl_self.cpp -o rccl_self -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("FAILED %s -> %d\n", #x, (int)e_); return 1; } } while (0)
int main()
{
    int dev = 0;
    ncclComm_t comm;
    CK(ncclCommInitAll(&comm, 1, &dev));
    const size_t n = 1 << 20;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; i++) h[i] = (float)i;
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int mode = 0; mode < 3; mode++) {
        CK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(b, 0, n * 4)); CK(hipDeviceSynchronize());
        hipStream_t ss = s1, rs = mode == 0 ? s1 : s2;
        CK(ncclGroupStart());
        if (mode == 2) { CK(ncclSend(a, n, ncclFloat, 0, comm, ss)); CK(ncclRecv(b, n, ncclFloat, 0, comm, rs)); }
        else { CK(ncclRecv(b, n, ncclFloat, 0, comm, rs)); CK(ncclSend(a, n, ncclFloat, 0, comm, ss)); }
        CK(ncclGroupEnd());
        CK(hipStreamSynchronize(rs)); 
        std::vector<float> g(n);
        CK(hipMemcpy(g.data(), b, n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < n; i++) bad += g[i] != h[i];
        CK(hipStreamSynchronize(ss));
        printf("mode %d (%s): %zu of %zu elements wrong after syncing the RECEIVE stream only\n", mode, mode == 0 ? "same stream" : (mode == 1 ? "recv first, two streams" : "send first, two streams"), bad, n);
    }
    ncclCommDestroy(comm);
    return 0;
}
