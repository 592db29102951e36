// Device math shared by the regression kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lws {

// exp(x) for x <= 0 from IEEE mul / fma / rint only (Cephes expf polynomial), so the result is a
// pure function of the float32 input on any IEEE machine; returns 0 below -80 (e^-80 ~ 1.8e-35).
__device__ __forceinline__ float lws_expf(float x)
{
    if (x < -80.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    int32_t bits = __float_as_int(y) + (((int32_t)n) << 23);
    return __int_as_float(bits);
}

// sum_k softmax_k(-c) * (start + k) over D values c[k*stride]: max-subtracted, S summed ascending,
// p_k = e_k / S (IEEE division), expectation summed ascending.  e_k is recomputed in the third
// pass instead of being kept in a D-sized register array (it is a pure function, so identical).
__device__ __forceinline__ float softargmin_pixel(const float *c, int64_t stride, int D, float start)
{
    float m = -c[0];
    for (int k = 1; k < D; ++k) m = fmaxf(m, -c[(int64_t)k * stride]);
    float S = 0.0f;
    for (int k = 0; k < D; ++k) S = S + lws_expf(-c[(int64_t)k * stride] - m);
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) {
        float pk = lws_expf(-c[(int64_t)k * stride] - m) / S;
        acc = acc + pk * (start + (float)k);
    }
    return acc;
}

}  // namespace lws
