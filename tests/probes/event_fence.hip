// What does an event record between two dependent kernels of one stream cost, with and without hipEventDisableSystemFence?
// (the backward pass hands work to its side streams with torch.cuda.Event records: the kernel trace shows 6-10 us of idle main
//  queue at each).  A 64-MB writer kernel (dirty L2 lines) followed by a record and the next kernel, 200 times.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void writer(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int main() {
    const size_t n = 16u << 20;
    float* p;
    hipMalloc(&p, n * 4);
    hipStream_t s, side;
    hipStreamCreate(&s);
    hipStreamCreate(&side);
    const unsigned flagsets[3] = {hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence, 0xFFFFFFFFu};
    const char* names[3] = {"DisableTiming", "DisableTiming|DisableSystemFence", "no event"};
    for (int rep = 0; rep < 2; ++rep)
    for (int f = 0; f < 3; ++f) {
        hipEvent_t ev[200];
        if (flagsets[f] != 0xFFFFFFFFu) for (auto& e : ev) hipEventCreateWithFlags(&e, flagsets[f]);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; ++i) {
            writer<<<1024, 256, 0, s>>>(p, n, (float)i);
            if (flagsets[f] != 0xFFFFFFFFu) {
                hipEventRecord(ev[i], s);
                hipStreamWaitEvent(side, ev[i], 0);
                writer<<<1, 64, 0, side>>>(p + n - 64, 64, 1.f);      // (a consumer on the side stream, as in the backward pass)
            }
        }
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("%-36s %8.2f us per iteration\n", names[f], us / 200);
        if (flagsets[f] != 0xFFFFFFFFu) for (auto& e : ev) hipEventDestroy(e);
    }
    return 0;
}
