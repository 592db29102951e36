// How many workgroups of 256 threads does a CU of the MI355X hold at once, as a function of the LDS each one asks for (and of its
// register count)?  Every workgroup records s_memrealtime at start and end and the CU it runs on (HW_ID / XCC_ID); the host
// reports the largest number of workgroups alive at once on one CU.  Behind the tile sizes of DESIGN.md section 4.
//   hipcc --offload-arch=gfx950 -O3 -o lds_residency tools/micro/lds_residency.hip && ./lds_residency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <vector>

template <int NREG>
__global__ __launch_bounds__(256) void k_spin(unsigned long long *rec, float *sink, int spin)
{
    extern __shared__ float lds[];
    unsigned long long t0, t1;
    unsigned hw, xcc;
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(t0), "=s"(hw), "=s"(xcc)::"memory");
    float r[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    for (int it = 0; it < spin; ++it) {
#pragma unroll
        for (int i = 0; i < NREG; ++i) r[i] = fmaf(r[i], 1.0001f, lds[(threadIdx.x + i) & 255]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 3 + 0] = t0;
        rec[blockIdx.x * 3 + 1] = t1;
        rec[blockIdx.x * 3 + 2] = ((unsigned long long)(xcc & 15u) << 32) | hw;
    }
}

template <int NREG>
static void run(int lds_bytes, int nwg, unsigned long long *drec, float *sink)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spin<NREG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_spin<NREG>, dim3(nwg), dim3(256), lds_bytes, 0, drec, sink, 2000 / NREG + 1);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("LDS %6d B: launch failed\n", lds_bytes);
        return;
    }
    std::vector<unsigned long long> rec(3 * (size_t)nwg);
    (void)hipMemcpy(rec.data(), drec, rec.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int i = 0; i < nwg; ++i) {
        const unsigned long long hw = rec[3 * i + 2];
        const unsigned long long cu = ((hw >> 32) & 15) * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 12) & 1) * 8 + ((hw >> 8) & 15);
        ev[cu].push_back({rec[3 * i], 1});
        ev[cu].push_back({rec[3 * i + 1], -1});
    }
    int peak = 0;
    for (auto &kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int n = 0;
        for (auto &e : kv.second) {
            n += e.second;
            peak = std::max(peak, n);
        }
    }
    unsigned long long a = ~0ull, b = 0;
    for (int i = 0; i < nwg; ++i) {
        a = std::min(a, rec[3 * i]);
        b = std::max(b, rec[3 * i + 1]);
    }
    printf("~%3d VGPRs, LDS %6d B per workgroup: CUs used %3zu, most workgroups alive at once on one CU %d, launch %.1f us\n", NREG + 16,
           lds_bytes, ev.size(), peak, (b - a) / 100.0);
}

int main()
{
    const int nwg = 256 * 8;
    unsigned long long *drec;
    float *sink;
    (void)hipMalloc(&drec, 3 * (size_t)nwg * 8);
    (void)hipMalloc(&sink, 4);
    for (int kb : {8, 16, 20, 26, 32, 36, 40, 41, 44, 48, 53, 54, 56, 64}) run<32>(kb * 1024, nwg, drec, sink);
    for (int b : {40448, 44288, 46080, 55296, 70656, 74880, 81920, 114688}) run<32>(b, nwg, drec, sink);
    for (int kb : {8, 32, 44}) run<96>(kb * 1024, nwg, drec, sink);
    for (int kb : {8, 32, 44}) run<120>(kb * 1024, nwg, drec, sink);
    return 0;
}
