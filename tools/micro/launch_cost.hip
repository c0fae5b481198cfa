// Host cost of the HIP calls the trunk engine makes per train step (one process, one GPU): kernel launches with a ~300-byte argument block
// on one stream, the same alternating between two streams, event record + cross-stream wait pairs, hipGetLastError, hipMemsetAsync.
// The device work is empty, so what is timed is the HOST side of each call (the queue never fills: a sync every 256 calls).
//   make -C tools/micro launch_cost.bin && tools/micro/launch_cost.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Args { char pad[304]; };
__global__ void empty_kernel(const Args a) { if (a.pad[0] == 77 && threadIdx.x == 1023) printf("x"); }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t ev[64];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    Args a = {};
    void* buf;
    hipMalloc(&buf, 1 << 20);
    const int N = 4096;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s0, a); if ((i & 255) == 255) hipStreamSynchronize(s0); }
        hipStreamSynchronize(s0);
        double t1 = now();
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s0, a); (void)hipGetLastError(); if ((i & 255) == 255) hipStreamSynchronize(s0); }
        hipStreamSynchronize(s0);
        double t2 = now();
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, (i & 1) ? s1 : s0, a); if ((i & 255) == 255) { hipStreamSynchronize(s0); hipStreamSynchronize(s1); } }
        hipStreamSynchronize(s0); hipStreamSynchronize(s1);
        double t3 = now();
        for (int i = 0; i < N; ++i) {   // launch on s0, record, s1 waits, launch on s1 (the engine's to_side)
            hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s0, a);
            hipEventRecord(ev[i & 63], s0);
            hipStreamWaitEvent(s1, ev[i & 63], 0);
            hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s1, a);
            if ((i & 63) == 63) { hipStreamSynchronize(s0); hipStreamSynchronize(s1); }
        }
        hipStreamSynchronize(s0); hipStreamSynchronize(s1);
        double t4 = now();
        for (int i = 0; i < N; ++i) { hipMemsetAsync(buf, 0, 4096, s0); if ((i & 255) == 255) hipStreamSynchronize(s0); }
        hipStreamSynchronize(s0);
        double t5 = now();
        if (rep == 1)
            printf("launch %.2f us | launch + hipGetLastError %.2f us | launch alternating two streams %.2f us | launch + record + wait + launch %.2f us | memsetAsync 4 KB %.2f us\n",
                   (t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6, (t3 - t2) / N * 1e6, (t4 - t3) / N * 1e6, (t5 - t4) / N * 1e6);
    }
    return 0;
}
