// Micro-test: is a kernel's output visible to the NEXT kernel's workgroups on OTHER XCDs when the runtime is told to skip the
// end-of-kernel release fence (DEBUG_CLR_SKIP_RELEASE_SCOPE=1)?  Round 5 found that switch worth +6 % on the batch-1 forward
// (profiles/r05/experiments/bench_env.txt); on a part whose 8 L2s are not coherent with each other a skipped write-back would
// mean STALE reads, so before the switch is even considered:
//   k_write   block b (XCD b % 8 under round-robin dispatch) writes val + index into its 256-word slice of a buffer that fits
//             in the L2s (so dirty lines stay there)
//   k_check   block b reads the slice of block b + shift (shift = 1, 3, 5: always a slice written on ANOTHER XCD) and counts
//             mismatches; val changes every iteration, so a stale line is the previous iteration's value
// in one stream, and with the check on a second stream behind an event (fork + join).  Also a read-modify-write chain
// (k_inc: every block increments a slice written by another XCD's block in the previous launch) -- the pattern of the
// Conv3D stack.  Prints the number of stale words per mode; run it with and without the switch:
//   hipcc --offload-arch=gfx950 -O3 -o release_scope tools/micro/release_scope.hip
//   ./release_scope;  DEBUG_CLR_SKIP_RELEASE_SCOPE=1 ./release_scope
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void k_write(uint32_t *buf, uint32_t val)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    buf[i] = val + i;
}

__global__ void k_check(const uint32_t *buf, uint32_t val, int shift, unsigned long long *errors)
{
    const uint32_t j = ((blockIdx.x + shift) % gridDim.x) * blockDim.x + threadIdx.x;
    if (buf[j] != val + j) atomicAdd(errors, 1ull);
}

// out slice b = in slice (b + shift) + 1: a chain of these is exact integer arithmetic, so the end value is known
__global__ void k_inc(const uint32_t *in, uint32_t *out, int shift)
{
    const uint32_t j = ((blockIdx.x + shift) % gridDim.x) * blockDim.x + threadIdx.x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = in[j] + 1u;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char *sw = getenv("DEBUG_CLR_SKIP_RELEASE_SCOPE");
    printf("DEBUG_CLR_SKIP_RELEASE_SCOPE=%s\n", sw ? sw : "(unset)");
    const int sizes_kb[] = {256, 2048, 16384, 131072};
    unsigned long long *err;
    CK(hipMalloc(&err, 8));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming | hipEventDisableSystemFence));
    for (int kb : sizes_kb) {
        const size_t words = (size_t)kb * 256;
        const int blocks = (int)(words / 256);
        uint32_t *buf, *buf2;
        CK(hipMalloc(&buf, words * 4));
        CK(hipMalloc(&buf2, words * 4));
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(err, 0, 8));
            CK(hipMemset(buf, 0, words * 4));
            CK(hipMemset(buf2, 0, words * 4));
            CK(hipDeviceSynchronize());
            const int iters = kb >= 16384 ? 200 : 2000;
            unsigned long long bad_chain = 0;
            if (mode == 0) {              // one stream: write, check, write, check ...
                for (int it = 0; it < iters; ++it) {
                    hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, s0, buf, (uint32_t)(it * 7919u));
                    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(256), 0, s0, buf, (uint32_t)(it * 7919u), 1 + 2 * (it % 3), err);
                }
            } else if (mode == 1) {       // two streams: write on s0, check on s1 behind an event; s0 waits for the check
                for (int it = 0; it < iters; ++it) {
                    if (it > 0) CK(hipStreamWaitEvent(s0, e1, 0));
                    hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, s0, buf, (uint32_t)(it * 104729u));
                    CK(hipEventRecord(e0, s0));
                    CK(hipStreamWaitEvent(s1, e0, 0));
                    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(256), 0, s1, buf, (uint32_t)(it * 104729u), 1 + 2 * (it % 3), err);
                    CK(hipEventRecord(e1, s1));
                }
            } else {                      // ping-pong chain of read-modify-write launches; the end value is iters
                for (int it = 0; it < iters; ++it)
                    hipLaunchKernelGGL(k_inc, dim3(blocks), dim3(256), 0, s0, (it & 1) ? buf2 : buf, (it & 1) ? buf : buf2, 1 + 2 * (it % 3));
                uint32_t *h = (uint32_t *)malloc(words * 4);
                CK(hipStreamSynchronize(s0));
                CK(hipMemcpy(h, (iters & 1) ? buf2 : buf, words * 4, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < words; ++i) bad_chain += h[i] != (uint32_t)iters;
                free(h);
            }
            CK(hipDeviceSynchronize());
            unsigned long long bad = 0;
            CK(hipMemcpy(&bad, err, 8, hipMemcpyDeviceToHost));
            const char *names[] = {"one stream write -> check", "two streams write -> event -> check", "one stream increment chain"};
            printf("%7d KB  %-38s %5d launches: %llu stale words\n", kb, names[mode], iters, mode == 2 ? bad_chain : bad);
        }
        CK(hipFree(buf));
        CK(hipFree(buf2));
    }
    return 0;
}
