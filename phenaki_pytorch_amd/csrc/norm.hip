// LayerNorm kernels (reference attention.py:29-36 gamma-only LayerNorm, nn.LayerNorm at
// attention.py:47 / cvivit.py:275-284; eps 1e-5, biased variance, f32 statistics).
// HBM-bound: algorithmic bytes = 4*D read + out_bytes*D written per row.
#include "common.hpp"

namespace pk {

// one wave per row, D % 4 == 0, D <= 64*4*VMAX
template <typename TO, int VMAX>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, int ldx,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, TO* __restrict__ out, int ldo,
                                                      float* __restrict__ out2, int ldo2, int M, int D,
                                                      int grp, int gstride, int goff) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    // optional output row remap: input row r -> (r / grp) * gstride + goff + r % grp  (frame-group concat)
    const int orow = grp > 0 ? (row / grp) * gstride + goff + row % grp : row;
    const float* xr = x + (size_t)row * ldx;
    const int nv = D >> 2;
    f32x4 v[VMAX];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = lane + i * 64;
        if (c < nv) { v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c * 4);
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (v[i][r] - mean) * rstd * gm[r];
            if (beta) y += *reinterpret_cast<const f32x4*>(beta + c * 4);
            if (out) store4(out + (size_t)orow * ldo + c * 4, y);
            if (out2) store4(out2 + (size_t)orow * ldo2 + c * 4, y);
        }
    }
}

}  // namespace pk
using namespace pk;

// out (f32 if out_kind == 0 else bf16) and/or out2 (always f32) receive LN(x) * gamma + beta.
// grp > 0 remaps output rows: row r -> (r / grp) * gstride + goff + r % grp (cvivit.py:549 frame concat).
extern "C" int pk_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                            void* out, int ldo, int out_kind, float* out2, int ldo2, int M, int D,
                            int grp, int gstride, int goff, void* stream) {
    if (M <= 0 || D <= 0 || !x || !gamma || (!out && !out2)) return PK_EINVAL;
    if ((D & 3) || (ldx & 3) || (out && (ldo & 3)) || (out2 && (ldo2 & 3))) return PK_EALIGN;
    if (D > 64 * 4 * 8) return PK_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((M + 3) / 4), block(256);
    const bool small = D <= 64 * 4 * 2;
    if (out_kind == 0) {
        if (small) hipLaunchKernelGGL((ln_rows_kernel<float, 2>), grid, block, 0, s, x, ldx, gamma, beta, eps, (float*)out, ldo, out2, ldo2, M, D, grp, gstride, goff);
        else hipLaunchKernelGGL((ln_rows_kernel<float, 8>), grid, block, 0, s, x, ldx, gamma, beta, eps, (float*)out, ldo, out2, ldo2, M, D, grp, gstride, goff);
    } else {
        if (small) hipLaunchKernelGGL((ln_rows_kernel<bf16, 2>), grid, block, 0, s, x, ldx, gamma, beta, eps, (bf16*)out, ldo, out2, ldo2, M, D, grp, gstride, goff);
        else hipLaunchKernelGGL((ln_rows_kernel<bf16, 8>), grid, block, 0, s, x, ldx, gamma, beta, eps, (bf16*)out, ldo, out2, ldo2, M, D, grp, gstride, goff);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}
