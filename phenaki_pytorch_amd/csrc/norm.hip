// LayerNorm kernels (reference attention.py:29-36 gamma-only LayerNorm, nn.LayerNorm at
// attention.py:47 / cvivit.py:275-284; eps 1e-5, biased variance, f32 statistics).
// HBM-bound: algorithmic bytes = 4*D read + out_bytes*D written per row.
#include "common.hpp"

namespace pk {

struct LnArgs {
    const float* x; int ldx;
    const float* gamma; const float* beta; float eps;
    void* out; int ldo;            // T (bf16 | f32 by the template) or null
    float* out2; int ldo2;         // f32 or null
    void* raw; int ldraw;          // optional: x itself converted to T (the un-normalised K/V source, attention.py:140-144)
    int M, D;
    int grp, gstride, goff;        // output row remap r -> (r / grp) * gstride + goff + r % grp        (grp > 0)
    int pb, pc;                    // output row remap (a, b, c) -> (a, c, b) over dims (M/(pb*pc), pb, pc) (pb > 0)
};

// one wave per row, D % 4 == 0, D <= 64*4*VMAX
template <typename TO, int VMAX>
__global__ __launch_bounds__(256) void ln_rows_kernel(const LnArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    int orow = row;
    if (p.grp > 0) orow = (row / p.grp) * p.gstride + p.goff + row % p.grp;
    else if (p.pb > 0) {
        const int c = row % p.pc, b = (row / p.pc) % p.pb, a = row / (p.pc * p.pb);
        orow = (a * p.pc + c) * p.pb + b;
    }
    const float* xr = p.x + (size_t)row * p.ldx;
    TO* out = reinterpret_cast<TO*>(p.out);
    TO* raw = reinterpret_cast<TO*>(p.raw);
    const int nv = p.D >> 2;
    // D <= 512 (VMAX = 2): gamma / beta are fetched together with x, before the statistics, so the normalise-and-store
    // loop has no load behind a store (9.0 -> 7.5 us on 9216 x 512); wider rows keep them in the store loop (registers)
    constexpr bool HOIST = VMAX <= 2;
    constexpr int NH = HOIST ? VMAX : 1;
    f32x4 v[VMAX], gm[NH], bt[NH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4);
            if constexpr (HOIST) {
                gm[i] = *reinterpret_cast<const f32x4*>(p.gamma + c * 4);
                bt[i] = p.beta ? *reinterpret_cast<const f32x4*>(p.beta + c * 4) : f32x4{0, 0, 0, 0};
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VMAX; ++i)
        if (lane + i * 64 < nv) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)p.D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)p.D + p.eps);
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            f32x4 g4, b4;
            if constexpr (HOIST) { g4 = gm[i]; b4 = bt[i]; }
            else {
                g4 = *reinterpret_cast<const f32x4*>(p.gamma + c * 4);
                b4 = p.beta ? *reinterpret_cast<const f32x4*>(p.beta + c * 4) : f32x4{0, 0, 0, 0};
            }
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (v[i][r] - mean) * rstd * g4[r] + b4[r];
            if (out) store4(out + (size_t)orow * p.ldo + c * 4, y);
            if (p.out2) store4(p.out2 + (size_t)orow * p.ldo2 + c * 4, y);
            if (raw) store4(raw + (size_t)orow * p.ldraw + c * 4, v[i]);
        }
    }
}

// one wave per row: out = x / max(||x||_2, 1e-12)
template <typename TO>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, int ldx, TO* __restrict__ out, int ldo, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        store4(out + (size_t)row * ldo + c, v * inv);
    }
}

}  // namespace pk
using namespace pk;

extern "C" int pk_l2norm_rows(const float* x, int ldx, void* out, int ldo, int out_kind, int M, int D, void* stream) {
    if (!x || !out || M <= 0 || D <= 0) return PK_EINVAL;
    if ((D & 3) || (ldx & 3) || (ldo & 3)) return PK_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((M + 3) / 4), block(256);
    if (out_kind == 0) hipLaunchKernelGGL((l2norm_rows_kernel<float>), grid, block, 0, s, x, ldx, (float*)out, ldo, M, D);
    else hipLaunchKernelGGL((l2norm_rows_kernel<bf16>), grid, block, 0, s, x, ldx, (bf16*)out, ldo, M, D);
    PK_CHECK_LAUNCH();
    return PK_OK;
}


// out / raw (f32 if out_kind == 0 else bf16) and/or out2 (always f32); out, out2 receive LN(x) * gamma + beta, raw
// receives x itself.  Output rows may be remapped: grp > 0: r -> (r / grp) * gstride + goff + r % grp (cvivit.py:549
// frame concat); pb > 0: rows seen as (a, b, c) with b < pb, c < pc are written at (a, c, b) -- the spatial <-> temporal
// 'b t (h w) <-> b (h w) t' rearranges of cvivit.py:468,472,488,496.
extern "C" int pk_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                            void* out, int ldo, int out_kind, float* out2, int ldo2, void* raw, int ldraw,
                            int M, int D, int grp, int gstride, int goff, int pb, int pc, void* stream) {
    if (M <= 0 || D <= 0 || !x || !gamma || (!out && !out2)) return PK_EINVAL;
    if ((D & 3) || (ldx & 3) || (out && (ldo & 3)) || (out2 && (ldo2 & 3)) || (raw && (ldraw & 3))) return PK_EALIGN;
    if (D > 64 * 4 * 8) return PK_EINVAL;
    if (pb > 0 && (pc <= 0 || M % (pb * pc))) return PK_EINVAL;
    if (grp > 0 && pb > 0) return PK_EINVAL;
    LnArgs p{x, ldx, gamma, beta, eps, out, ldo, out2, ldo2, raw, ldraw, M, D, grp, gstride, goff, pb, pc};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((M + 3) / 4), block(256);
    const bool small = D <= 64 * 4 * 2;
    if (out_kind == 0) {
        if (small) hipLaunchKernelGGL((ln_rows_kernel<float, 2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((ln_rows_kernel<float, 8>), grid, block, 0, s, p);
    } else {
        if (small) hipLaunchKernelGGL((ln_rows_kernel<bf16, 2>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((ln_rows_kernel<bf16, 8>), grid, block, 0, s, p);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}
