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

// Final LayerNorm of the C-ViViT encoder fused with the lookup-free quantizer (cvivit.py:472 norm_out -> :570 LFQ): one wave
// per row normalises it in registers and takes the cd <= 16 dot products with project_in, so the (M, D) f32 token matrix is
// neither written nor read back and one launch disappears.  The 16 wave-wide sums are folded with a halving butterfly (the
// lanes exchange the half of the partials they do not keep: 8 + 4 + 2 + 1 + 1 + 1 = 17 shuffles instead of 16 x 6); lane
// group k = lane >> 2 ends up with sum k, one ballot collects the sign bits.  ids[orow] = sum_k (proj_k > 0) << (cd-1-k),
// orow = the (a, b, c) -> (a, c, b) row permutation of pk_layernorm.  tokens (f32, optional) receives LN(x) for callers that
// still need it; proj (optional) the pre-sign values for the parity margin audit.
template <int VMAX, int R>
__global__ __launch_bounds__(256) void ln_lfq_kernel(const LnArgs p, const float* __restrict__ wp, const float* __restrict__ bp,
                                                      int cd, long long* __restrict__ ids, float* __restrict__ proj) {
    // one wave owns R consecutive rows: every project_in vector it loads is applied to R rows (the first version re-read the
    // 32 KB of project_in per ROW through L1: 10 us for 4608 rows)
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= p.M) return;
    const int nv = p.D >> 2;
    f32x4 y[R][VMAX];
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r < p.M ? row0 + r : p.M - 1;          // tail rows recompute the last row, their ids are not stored
        const float* xr = p.x + (size_t)row * p.ldx;
#pragma unroll
        for (int i = 0; i < VMAX; ++i) {
            const int c = lane + i * 64;
            y[r][i] = c < nv ? *reinterpret_cast<const f32x4*>(xr + c * 4) : f32x4{0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VMAX; ++i) s += (y[r][i][0] + y[r][i][1]) + (y[r][i][2] + y[r][i][3]);
        mean[r] = wave_sum(s) / (float)p.D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VMAX; ++i)
            if (lane + i * 64 < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = y[r][i][e] - mean[r]; q += d * d; }
            }
        rstd[r] = 1.0f / sqrtf(wave_sum(q) / (float)p.D + p.eps);
    }
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.gamma + c * 4);
            const f32x4 b4 = p.beta ? *reinterpret_cast<const f32x4*>(p.beta + c * 4) : f32x4{0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[r][i][e] = (y[r][i][e] - mean[r]) * rstd[r] * g4[e] + b4[e];
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) y[r][i] = f32x4{0, 0, 0, 0};
        }
    }
    float part[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int k = 0; k < 16; ++k) part[r][k] = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < cd) {
#pragma unroll
            for (int i = 0; i < VMAX; ++i) {
                const int c = lane + i * 64;
                const f32x4 wv = c < nv ? *reinterpret_cast<const f32x4*>(wp + (size_t)k * p.D + c * 4) : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < R; ++r)
                    part[r][k] += (y[r][i][0] * wv[0] + y[r][i][1] * wv[1]) + (y[r][i][2] * wv[2] + y[r][i][3] * wv[3]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row >= p.M) break;                               // wave-uniform
        int orow = row;
        if (p.pb > 0) {
            const int c = row % p.pc, b = (row / p.pc) % p.pb, a = row / (p.pc * p.pb);
            orow = (a * p.pc + c) * p.pb + b;
        }
        if (p.out2) {
#pragma unroll
            for (int i = 0; i < VMAX; ++i)
                if (lane + i * 64 < nv) store4(p.out2 + (size_t)orow * p.ldo2 + (lane + i * 64) * 4, y[r][i]);
        }
        // halving butterfly: after the step with lane bit `bit`, a lane keeps the partials whose index has that bit equal to its own
        float h8[8], h4[4], h2[2], h1;
        {
            const bool hi = lane & 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float keep = hi ? part[r][i + 8] : part[r][i], send = hi ? part[r][i] : part[r][i + 8];
                h8[i] = keep + __shfl_xor(send, 32, 64);
            }
        }
        {
            const bool hi = lane & 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float keep = hi ? h8[i + 4] : h8[i], send = hi ? h8[i] : h8[i + 4];
                h4[i] = keep + __shfl_xor(send, 16, 64);
            }
        }
        {
            const bool hi = lane & 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float keep = hi ? h4[i + 2] : h4[i], send = hi ? h4[i] : h4[i + 2];
                h2[i] = keep + __shfl_xor(send, 8, 64);
            }
        }
        {
            const bool hi = lane & 4;
            const float keep = hi ? h2[1] : h2[0], send = hi ? h2[0] : h2[1];
            h1 = keep + __shfl_xor(send, 4, 64);
        }
        h1 += __shfl_xor(h1, 2, 64);
        h1 += __shfl_xor(h1, 1, 64);
        const int k = lane >> 2;                                 // bits 5..2 of the lane = bits 3..0 of k
        const float val = h1 + (k < cd ? bp[k] : 0.f);
        if (proj && k < cd && (lane & 3) == 0) proj[(size_t)orow * cd + k] = val;
        const unsigned long long bal = __ballot(val > 0.f);
        if (lane == 0) {
            long long id = 0;
            for (int kk = 0; kk < cd; ++kk) id |= (long long)((bal >> (4 * kk)) & 1ull) << (cd - 1 - kk);
            ids[orow] = id;
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

// T5LayerNorm (HuggingFace transformers modeling_t5.py T5LayerNorm, the text encoder the reference calls in t5.py:64-103): no mean
// subtraction, no bias -- y = x * rsqrt(mean(x^2) + eps) * w, statistics in f32.  rowmask (or null): rows with rowmask[row] == 0 are
// written as zeros (t5.py:97-100 masked_fill of the padded positions, fused into the encoder's final norm).  One wave per row.
template <typename TO>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, float eps,
                                                           const unsigned char* __restrict__ rowmask, TO* __restrict__ out, int ldo, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    const bool keep = !rowmask || rowmask[row] != 0;
    float q = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c), g = *reinterpret_cast<const f32x4*>(w + c);
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = keep ? v[r] * rstd * g[r] : 0.f;
        store4(out + (size_t)row * ldo + c, y);
    }
}

}  // namespace pk
using namespace pk;

extern "C" int pk_rmsnorm(const float* x, int ldx, const float* w, float eps, const unsigned char* rowmask, void* out, int ldo,
                          int out_kind, int M, int D, void* stream) {
    if (!x || !w || !out || M <= 0 || D <= 0) return PK_EINVAL;
    if ((D & 3) || (ldx & 3) || (ldo & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15)) return PK_EALIGN;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((M + 3) / 4), block(256);
    if (out_kind == 0) hipLaunchKernelGGL((rmsnorm_rows_kernel<float>), grid, block, 0, s, x, ldx, w, eps, rowmask, (float*)out, ldo, M, D);
    else hipLaunchKernelGGL((rmsnorm_rows_kernel<bf16>), grid, block, 0, s, x, ldx, w, eps, rowmask, (bf16*)out, ldo, M, D);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

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


// ids[orow] <- LFQ(LayerNorm(x[row])) (see ln_lfq_kernel); cd <= 16, D <= 2048; tokens / proj optional outputs; pb, pc as in
// pk_layernorm (orow = the transposed row), pb = 0: orow = row
extern "C" int pk_layernorm_lfq(const float* x, int ldx, const float* gamma, const float* beta, float eps, const float* wp,
                                const float* bp, long long* ids, float* tokens, int ldt, float* proj, int M, int D, int cd,
                                int pb, int pc, void* stream) {
    if (M <= 0 || D <= 0 || !x || !gamma || !wp || !bp || !ids || cd <= 0 || cd > 16) return PK_EINVAL;
    if ((D & 3) || (ldx & 3) || (tokens && (ldt & 3))) return PK_EALIGN;
    if (D > 64 * 4 * 8) return PK_EINVAL;
    if (pb > 0 && (pc <= 0 || M % (pb * pc))) return PK_EINVAL;
    LnArgs p{x, ldx, gamma, beta, eps, nullptr, 0, tokens, ldt, nullptr, 0, M, D, 0, 0, 0, pb, pc};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 block(256);
    // rows per wave: 1 measured 10.4 us, 4 measured 14.1 us at 4608 x 512 (fewer, longer waves lose more than the 4x fewer
    // project_in reads win); 2 is the compromise kept for large M only
    if (D <= 64 * 4 * 2 && M >= 32768) hipLaunchKernelGGL((ln_lfq_kernel<2, 2>), dim3((M + 7) / 8), block, 0, s, p, wp, bp, cd, ids, proj);
    else if (D <= 64 * 4 * 2) hipLaunchKernelGGL((ln_lfq_kernel<2, 1>), dim3((M + 3) / 4), block, 0, s, p, wp, bp, cd, ids, proj);
    else hipLaunchKernelGGL((ln_lfq_kernel<8, 1>), dim3((M + 3) / 4), block, 0, s, p, wp, bp, cd, ids, proj);
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
