// probe (round 5): does RCCL accept TWO communicators (ranks 0 and 1 of one clique) on ONE device, so that
// the grouped ncclSend / ncclRecv of vr_group's gather can run between distinct ranks on a one-GPU box?
//   hipcc rccl_two_on_one.cpp -o rccl_two_on_one -lrccl      (tries ncclCommInitAll with devices {0, 0}, then ncclCommInitRank)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <thread>
#include <vector>
int main()
{
    int devs[2] = {0, 0};
    ncclComm_t comms[2] = {nullptr, nullptr};
    ncclResult_t r = ncclCommInitAll(comms, 2, devs);
    printf("ncclCommInitAll(2 ranks on device 0): %s\n", ncclGetErrorString(r));
    if (r != ncclSuccess) {
        ncclUniqueId id;
        r = ncclGetUniqueId(&id);
        printf("ncclGetUniqueId: %s\n", ncclGetErrorString(r));
        ncclResult_t rr[2] = {ncclSuccess, ncclSuccess};
        std::thread t([&] { hipSetDevice(0); rr[1] = ncclCommInitRank(&comms[1], 2, id, 1); });
        hipSetDevice(0);
        rr[0] = ncclCommInitRank(&comms[0], 2, id, 0);
        t.join();
        printf("ncclCommInitRank x 2 on device 0: rank 0 %s, rank 1 %s\n", ncclGetErrorString(rr[0]), ncclGetErrorString(rr[1]));
        if (rr[0] != ncclSuccess || rr[1] != ncclSuccess) { printf("RESULT: RCCL refuses two ranks on one device: the gather between distinct ranks cannot be rehearsed on this box\n"); return 0; }
    }
    const size_t n = 1 << 20;
    float *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; i++) h[i] = (float)i;
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(b, 0, n * 4);
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    ncclGroupStart();
    ncclRecv(b, n, ncclFloat, 1, comms[0], s0);      // the root's receive from rank 1
    ncclSend(a, n, ncclFloat, 0, comms[1], s1);      // rank 1's send to the root
    r = ncclGroupEnd();
    printf("grouped recv(rank 0) + send(rank 1): %s\n", ncclGetErrorString(r));
    hipStreamSynchronize(s0); hipStreamSynchronize(s1);
    std::vector<float> g(n);
    hipMemcpy(g.data(), b, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < n; i++) bad += g[i] != h[i];
    printf("RESULT: %zu of %zu elements wrong\n", bad, n);
    return 0;
}
