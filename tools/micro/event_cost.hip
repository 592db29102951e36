// Micro-benchmark: what does a cross-stream hand-off cost the stream that carries the critical chain?
// lws_forward's batch-1 chain has two hipEventRecord and three hipStreamWaitEvent on the caller's stream (side-stream forks and
// joins); the kernel trace of round 4 shows ~5 us of idle queue at each.  Priced here on a chain of N short dependent kernels:
//   base      N kernels back to back on s0
//   record    + hipEventRecord(e_i, s0) after every kernel (nobody waits)
//   fork      + record on s0, hipStreamWaitEvent(s1, e_i), a short kernel on s1 (the side stream really consumes it)
//   join      a short kernel + hipEventRecord(f_i) on s1 issued FIRST (complete long before), hipStreamWaitEvent(s0, f_i) before kernel i
//   wrval     hipStreamWriteValue32(s0, flag_i) after every kernel instead of the event record
//   waitval   s1 writes flag_i early; hipStreamWaitValue32(s0, flag_i, ==) before kernel i instead of the event wait
// Round 5 (the host of lws_forward runs a whole forward AHEAD of the device, so at hipStreamWaitEvent time nothing is complete):
//   extstop   the event bound to the kernel's own completion signal: hipExtLaunchKernelGGL(..., stopEvent = e_i) -- no marker packet
//   extfork   extstop + hipStreamWaitEvent(s1, e_i) + a short kernel on s1
//   joinlive  fork, and kernel i of s0 waits for the short s1 kernel that was forked after kernel i-2 (f_{i-2}): the event is
//             NOT complete when the host calls hipStreamWaitEvent (a barrier packet is really queued) but IS when the command
//             processor reaches it -- what lws_forward's three joins look like; its cost = joinlive - fork
//   extjoin   extfork + the same joins
// Output: us per kernel of the chain and the extra us per hand-off against `base`.
//   hipcc --offload-arch=gfx950 -O3 -o event_cost tools/micro/event_cost.hip && ./event_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>

__global__ void k_short(float *p, int n)
{
    float x = p[threadIdx.x];
    for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;
    p[threadIdx.x] = x;
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main()
{
    const int N = 40, REP = 30, SPIN = 1500;
    float *a, *b;
    CK(hipMalloc(&a, 4096));
    CK(hipMalloc(&b, 4096));
    CK(hipMemset(a, 0, 4096));
    CK(hipMemset(b, 0, 4096));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(N), fv(N);
    for (int i = 0; i < N; ++i) {
        CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventDisableSystemFence));
        CK(hipEventCreateWithFlags(&fv[i], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    uint32_t *flags = nullptr;
    const bool have_val = hipExtMallocWithFlags((void **)&flags, N * 4 * 2, hipMallocSignalMemory) == hipSuccess;
    if (!have_val) {
        (void)hipGetLastError();
        printf("(hipMallocSignalMemory unavailable: wrval / waitval skipped)\n");
    }
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    const char *names[] = {"base", "record", "fork", "join", "wrval", "waitval", "extstop", "extfork", "joinlive", "extjoin"};
    double base_us = 0;
    for (int mode = 0; mode < 10; ++mode) {
        if ((mode == 4 || mode == 5) && !have_val) continue;
        float best = 1e30f;
        for (int rep = 0; rep < REP; ++rep) {
            if (have_val) CK(hipMemset(flags, 0, N * 4 * 2));
            CK(hipDeviceSynchronize());
            if (mode == 3)
                for (int i = 0; i < N; ++i) {              // the side stream's work, complete long before s0 gets there
                    hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, s1, b, 10);
                    CK(hipEventRecord(fv[i], s1));
                }
            if (mode == 5)
                for (int i = 0; i < N; ++i) CK(hipStreamWriteValue32(s1, flags + i, 1u, 0));
            CK(hipEventRecord(t0, s0));
            for (int i = 0; i < N; ++i) {
                if (mode == 3) CK(hipStreamWaitEvent(s0, fv[i], 0));
                if (mode == 5) CK(hipStreamWaitValue32(s0, flags + i, 1u, hipStreamWaitValueEq, 0xffffffffu));
                if ((mode == 8 || mode == 9) && i >= 2) CK(hipStreamWaitEvent(s0, fv[i - 2], 0));
                if (mode == 6 || mode == 7 || mode == 9)
                    hipExtLaunchKernelGGL(k_short, dim3(256), dim3(64), 0, s0, nullptr, ev[i], 0, a, SPIN);
                else
                    hipLaunchKernelGGL(k_short, dim3(256), dim3(64), 0, s0, a, SPIN);
                if (mode == 1 || mode == 2 || mode == 8) CK(hipEventRecord(ev[i], s0));
                if (mode == 2 || mode == 7 || mode == 8 || mode == 9) {
                    CK(hipStreamWaitEvent(s1, ev[i], 0));
                    hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, s1, b, 10);
                    if (mode == 8 || mode == 9) CK(hipEventRecord(fv[i], s1));
                }
                if (mode == 4) CK(hipStreamWriteValue32(s0, flags + N + i, 1u, 0));
            }
            CK(hipEventRecord(t1, s0));
            CK(hipEventSynchronize(t1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (ms < best) best = ms;
        }
        const double us = 1e3 * best / N;
        if (mode == 0) base_us = us;
        printf("%-8s %7.2f us per kernel of the chain  (+%.2f us per hand-off)\n", names[mode], us, us - base_us);
    }
    return 0;
}
