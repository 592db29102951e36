// Micro-test: does hipExtAnyOrderLaunch (a dispatch packet without the barrier bit) let a short kernel overlap the kernel queued
// BEFORE it in the same stream on gfx950?  (hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for
// hipExtModuleLaunchKernel.)  If it did, lws_forward's side branch could live in the caller's stream with no events at all: a
// side kernel queued any-order behind chain kernel K_j runs beside it, and K_j+1's barrier bit is the join.
//   chain      N x [long kernel]                                   -> t0 per step
//   inorder    N x [long kernel][short kernel]                     -> t0 + t_short
//   anyorder   N x [long kernel][short kernel, any-order flag]     -> t0 if the flag works, t0 + t_short if it is ignored
//   hipcc --offload-arch=gfx950 -O3 -o anyorder tools/micro/anyorder.hip && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>

__global__ void k_spin(float *p, int n)
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
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int N = 40, REP = 20;
    float *a, *b;
    CK(hipMalloc(&a, 4096));
    CK(hipMalloc(&b, 4096));
    CK(hipMemset(a, 0, 4096));
    CK(hipMemset(b, 0, 4096));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));
    const char *names[] = {"chain", "inorder", "anyorder"};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < REP; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, s));
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, s, a, 1500);
                if (mode == 1) hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, b, 500);
                if (mode == 2) hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, b, 500);
            }
            CK(hipEventRecord(t1, s));
            CK(hipEventSynchronize(t1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (ms < best) best = ms;
        }
        printf("%-9s %7.2f us per step\n", names[mode], 1e3 * best / N);
    }
    return 0;
}
