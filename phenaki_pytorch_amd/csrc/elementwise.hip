// Small HBM-bound kernels of the hot path: PEG depthwise conv, LFQ encode/decode, token+position
// embedding, continuous-position-bias input layer, critic score head.
#include "common.hpp"

namespace pk {

// ---- PEG (reference attention.py:57-85): depthwise 3x3x3 conv over a channels-last (B,T,H,W,D) view
// of the token buffer, zero padding, time pad (2,0) if causal else (1,1), bias, + residual (attention.py:323).
// wt is the conv weight pre-transposed to [27][D] (tap-major) so channel loads are 16-byte vectors.
__global__ __launch_bounds__(256) void peg_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                  const float* __restrict__ bias, float* __restrict__ out, bf16* __restrict__ out_t,
                                                  int B, int T, int H, int W, int D, int tfront, long total_vec) {
    const int dv = D >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total_vec; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % dv) * 4;
        long p = idx / dv;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H); p /= H;
        const int t = (int)(p % T); const int b = (int)(p / T);
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
        f32x4 center = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int ts = t + dt - tfront;
            if (ts < 0 || ts >= T) continue;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const int hs = h + dh - 1;
                if (hs < 0 || hs >= H) continue;
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const int ws = w + dw - 1;
                    if (ws < 0 || ws >= W) continue;
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((((size_t)b * T + ts) * H + hs) * W + ws) * D + c);
                    const f32x4 kv = *reinterpret_cast<const f32x4*>(wt + (size_t)((dt * 3 + dh) * 3 + dw) * D + c);
                    acc += xv * kv;
                    if (ts == t && hs == h && ws == w) center = xv;
                }
            }
        }
        const size_t o = ((((size_t)b * T + t) * H + h) * W + w) * D + c;
        *reinterpret_cast<f32x4*>(out + o) = acc + center;
        if (out_t) store4(out_t + o, acc + center);
    }
}

// same stencil, one thread per (b, t, h, 4 channels) producing the whole W-row: every input row segment is loaded once
// and reused by its 3 taps, and the 27 weight vectors are loaded once per thread (3x fewer L2 reads than peg_kernel)
// flip (round 6): the ADJOINT stencil for the backward pass, dx = dy + conv^T(dy) -- tap (dt, dh, dw) reads weight 26 - tap (all three axes mirrored) and the
// caller passes tfront' = 2 - tfront and no bias: the same row-reusing loop instead of the training step's own 27-gather kernel (23 -> 10 us)
template <int WW>
__global__ __launch_bounds__(256) void peg_row_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                      const float* __restrict__ bias, float* __restrict__ out, bf16* __restrict__ out_t,
                                                      int B, int T, int H, int D, int tfront, long total, int flip = 0) {
    const int dv = D >> 2;
    // XCD-contiguous order (common.hpp): a stencil row's 9 neighbour rows then sit in the SAME XCD's L2 -- 19.2 -> 13.3 us
    // at (16,9,8,8,512), bit-identical output (tools/peg_bench.hip)
    const long idx = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % dv) * 4;
    long p = idx / dv;
    const int h = (int)(p % H); p /= H;
    const int t = (int)(p % T); const int b = (int)(p / T);
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[WW];
#pragma unroll
    for (int w = 0; w < WW; ++w) acc[w] = bv;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
        const int ts = t + dt - tfront;
        if (ts < 0 || ts >= T) continue;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hs = h + dh - 1;
            if (hs < 0 || hs >= H) continue;
            const float* row = x + (((size_t)b * T + ts) * H + hs) * WW * D + c;
            f32x4 xr[WW];
#pragma unroll
            for (int w = 0; w < WW; ++w) xr[w] = *reinterpret_cast<const f32x4*>(row + (size_t)w * D);
            const int tb = (dt * 3 + dh) * 3;
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(wt + (size_t)(flip ? 26 - tb : tb) * D + c);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(wt + (size_t)(flip ? 25 - tb : tb + 1) * D + c);
            const f32x4 k2 = *reinterpret_cast<const f32x4*>(wt + (size_t)(flip ? 24 - tb : tb + 2) * D + c);
#pragma unroll
            for (int w = 0; w < WW; ++w) {
                if (w > 0) acc[w] += xr[w - 1] * k0;
                acc[w] += xr[w] * k1;
                if (w + 1 < WW) acc[w] += xr[w + 1] * k2;
                if (dt == tfront && dh == 1) acc[w] += xr[w];            // + residual (attention.py:323)
            }
        }
    }
    const size_t o0 = (((size_t)b * T + t) * H + h) * WW * D + c;
#pragma unroll
    for (int w = 0; w < WW; ++w) *reinterpret_cast<f32x4*>(out + o0 + (size_t)w * D) = acc[w];
    if (out_t) {                                             // bf16 copy: the next GEMM's LDS-DMA operand
#pragma unroll
        for (int w = 0; w < WW; ++w) store4(out_t + o0 + (size_t)w * D, acc[w]);
    }
}

// (round 5: an LDS-staged form -- one sequence x one 16-channel slice per workgroup, the T*H*W x 64 B slab loaded once, every thread walking the time
//  axis of its (h, w) column -- was built, measured and removed: 12.9 vs 9.4 us at (8,9,8,8,512), 16.5 vs 15.0 at 16 sequences; it wins only from
//  64 sequences on (43 vs 48.5 us).  256 workgroups of 4 waves, each waiting on its own 37 KB, are latency-bound where peg_row_kernel's 1152
//  workgroups are not: the 9x re-read it saves was never the bound.  profiles/peg_slab_r05.txt)
// ---- LFQ (vector-quantize-pytorch LFQ restated in oracle/lfq.py; call sites cvivit.py:570, :439)
// encode: proj = x @ Wp^T + bp (f32, one wave per token), ids = sum_k (proj_k > 0) << (cd-1-k)  (MSB first)
template <int CD>
__global__ __launch_bounds__(256) void lfq_encode_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ wp,
                                                         const float* __restrict__ bp, long long* __restrict__ ids,
                                                         float* __restrict__ proj, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float part[CD];
#pragma unroll
    for (int k = 0; k < CD; ++k) part[k] = 0.f;
    const float* xr = x + (size_t)row * ldx;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
        for (int k = 0; k < CD; ++k) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (size_t)k * D + c);
            part[k] += (xv[0] * wv[0] + xv[1] * wv[1]) + (xv[2] * wv[2] + xv[3] * wv[3]);
        }
    }
    long long id = 0;
#pragma unroll
    for (int k = 0; k < CD; ++k) {
        const float v = wave_sum(part[k]) + bp[k];
        if (proj && lane == 0) proj[(size_t)row * CD + k] = v;
        id |= (long long)(v > 0.f ? 1 : 0) << (CD - 1 - k);
    }
    if (lane == 0) ids[row] = id;
}

// decode: codes = (+1 | -1 per bit, MSB first) @ Wo^T + bo ; wo is [D][cd].
// A rank-cd outer product per row: each thread keeps the cd weights (+ bias) of its 4 output channels in registers for the
// whole launch and walks the rows (grid-stride), so a row costs one 8-byte id read (broadcast), 4*cd signed adds and ONE
// 16-byte store per thread -- the 2 KB write per token is the only HBM traffic (the first version re-read wo per
// element: 74 us for 9.4 MB).  HBM-bound: algorithmic bytes = 4*D + 8 per row.
// token id of flat row `row` over (b, i), i < n_prime + n: from ids_prime[b][i] for the primed positions, ids[b][i - n_prime] otherwise
// (no torch.cat of the two id arrays on the host); ids_prime == NULL: ids[row]
__device__ __forceinline__ long long lfq_row_id(const long long* __restrict__ ids_prime, int n_prime, const long long* __restrict__ ids, int n, int row) {
    if (!ids_prime) return ids[row];
    const int n_tot = n_prime + n, b = row / n_tot, i = row - b * n_tot;
    return i < n_prime ? ids_prime[(size_t)b * n_prime + i] : ids[(size_t)b * n + (i - n_prime)];
}
// output row of input row r = (a, b, c) with extents (*, pb, pc): (a, c, b) -- the decoder's temporal transformer wants its rows as
// '(b h w) t' (cvivit.py:482), so the transposed order is written directly instead of through a transpose().contiguous() pass
__device__ __forceinline__ int lfq_out_row(int r, int pb, int pc) {
    if (pb <= 0) return r;
    const int c = r % pc, ab = r / pc, b = ab % pb, a = ab / pb;
    return (a * pc + c) * pb + b;
}

template <int CD>
__global__ __launch_bounds__(256) void lfq_decode_kernel(const long long* __restrict__ ids_prime, int n_prime, const long long* __restrict__ ids, int n,
                                                         const float* __restrict__ wo,
                                                         const float* __restrict__ bo, float* __restrict__ out,
                                                         int M, int D, int row_stride, int pb, int pc) {
    const int dv = D >> 2;                                   // 16-byte column groups per row
    const int lanes = blockDim.x;                            // host: lanes == dv * rpb, rpb rows in flight per block
    const int cg = threadIdx.x % dv, rl = threadIdx.x / dv;
    // this thread's 4 output channels x CD weights are 4*CD CONTIGUOUS floats of wo ([D][CD] row-major): 16-byte loads
    float w[4][CD];
    f32x4 b4 = *reinterpret_cast<const f32x4*>(bo + cg * 4);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wo + (size_t)cg * 4 * CD);
#pragma unroll
        for (int q = 0; q < CD; ++q) {
            const f32x4 t = src[q];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[(q * 4 + r) / CD][(q * 4 + r) % CD] = t[r];
        }
    }
    for (int row = blockIdx.x * (lanes / dv) + rl; row < M; row += row_stride) {
        const unsigned long long id = (unsigned long long)lfq_row_id(ids_prime, n_prime, ids, n, row);
        f32x4 acc = b4;
#pragma unroll
        for (int k = 0; k < CD; ++k) {
            const bool on = (id >> (CD - 1 - k)) & 1ull;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += on ? w[e][k] : -w[e][k];
        }
        *reinterpret_cast<f32x4*>(out + (size_t)lfq_out_row(row, pb, pc) * D + cg * 4) = acc;
    }
}

// generic fallback (any D, cd <= 62): one thread per output element
__global__ __launch_bounds__(256) void lfq_decode_generic_kernel(const long long* __restrict__ ids_prime, int n_prime, const long long* __restrict__ ids, int n,
                                                                 const float* __restrict__ wo,
                                                                 const float* __restrict__ bo, float* __restrict__ out,
                                                                 int M, int D, int cd, int pb, int pc) {
    const long total = (long)M * D;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int d = (int)(idx % D);
        const int row = (int)(idx / D);
        const long long id = lfq_row_id(ids_prime, n_prime, ids, n, row);
        float acc = bo[d];
        for (int k = 0; k < cd; ++k) {
            const float sgn = ((id >> (cd - 1 - k)) & 1) ? 1.f : -1.f;
            acc += sgn * wo[(size_t)d * cd + k];
        }
        out[(size_t)lfq_out_row(row, pb, pc) * D + d] = acc;
    }
}

// ---- token + position embedding (reference phenaki_pytorch.py:194-197, :290-291)
// Row r = s * n_tot + i of the output is sequence s, position i.  The token id comes from ids_prime[b][i] for i < n_prime
// (the primed frames' tokens, phenaki_pytorch.py:500) and from ids[b][i - n_prime] otherwise, with b = s % nb: the S = 2*nb
// sequences of a classifier-free-guidance batch (cond | null) read the SAME nb id rows -- no torch.cat on the host.
__global__ __launch_bounds__(256) void embed_kernel(const long long* __restrict__ ids_prime, int n_prime, const long long* __restrict__ ids,
                                                    int n, int nb, const float* __restrict__ tok, const float* __restrict__ pos,
                                                    float* __restrict__ out, bf16* __restrict__ out_t, int D, long total_vec) {
    const int dv = D >> 2;
    const int n_tot = n_prime + n;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total_vec; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % dv) * 4;
        const long r = idx / dv;
        const int i = (int)(r % n_tot);
        const int b = (int)((r / n_tot) % nb);
        const long long id = i < n_prime ? ids_prime[(size_t)b * n_prime + i] : ids[(size_t)b * n + (i - n_prime)];
        const f32x4 a = *reinterpret_cast<const f32x4*>(pos + (size_t)i * D + c);
        const f32x4 t = *reinterpret_cast<const f32x4*>(tok + (size_t)id * D + c);
        *reinterpret_cast<f32x4*>(out + (size_t)r * D + c) = a + t;
        if (out_t) store4(out_t + (size_t)r * D + c, a + t);
    }
}

// ---- ContinuousPositionBias input layer (reference attention.py:257-272): for every (i, j) of the
// flattened 'ij' meshgrid, rel = pos_i - pos_j, v = sign(rel) * log(|rel| + 1), h = leaky_relu(W0 v + b0, 0.1)
__global__ __launch_bounds__(256) void cpb_input_kernel(const float* __restrict__ w0, const float* __restrict__ b0,
                                                        float* __restrict__ out, int d0, int d1, int d2, int nd, int D, long total) {
    const int n = d0 * d1 * d2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % D);
        const long pr = idx / D;
        const int j = (int)(pr % n), i = (int)(pr / n);
        int pi[3] = {i / (d1 * d2), (i / d2) % d1, i % d2};
        int pj[3] = {j / (d1 * d2), (j / d2) % d1, j % d2};
        float acc = b0[c];
        for (int a = 0; a < nd; ++a) {
            const int q = 3 - nd + a;                      // nd == 2 uses the trailing (h, w) axes
            const float rel = (float)(pi[q] - pj[q]);
            const float sg = rel > 0.f ? 1.f : (rel < 0.f ? -1.f : 0.f);
            acc += w0[(size_t)c * nd + a] * (sg * logf(fabsf(rel) + 1.f));
        }
        out[idx] = acc > 0.f ? acc : 0.1f * acc;
    }
}

// ---- critic score head (reference phenaki_pytorch.py:246-263, 523-545):
//   s[r] = x[r] . w + b ;  out[b][i] = null + (cond - null) * scale  + noise_mult * (u - 0.5)
// rows are laid out [cond sequences | null sequences]; prime positions are dropped (i >= n_prime).
__global__ __launch_bounds__(256) void critic_head_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                          const float* __restrict__ bptr, int D, int nb, int n_tot, int n_prime, int has_null,
                                                          float scale, const float* __restrict__ u, float noise_mult,
                                                          unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                          float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int n = n_tot - n_prime;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);     // r indexes (batch, i) of the OUTPUT
    if (r >= nb * n) return;
    const int bi = r / n, i = r % n + n_prime;
    const float b = bptr ? bptr[0] : 0.f;
    auto dot = [&](const float* xr) {
        float s = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
            s += (xv[0] * wv[0] + xv[1] * wv[1]) + (xv[2] * wv[2] + xv[3] * wv[3]);
        }
        return wave_sum(s) + b;
    };
    float s = dot(x + ((size_t)bi * n_tot + i) * ldx);
    if (has_null) {
        const float sn = dot(x + ((size_t)(nb + bi) * n_tot + i) * ldx);
        s = sn + (s - sn) * scale;
    }
    if (u) s += noise_mult * (u[r] - 0.5f);
    else if (noise_mult != 0.f) {                           // FAST mode: the uniform draw of phenaki_pytorch.py:541 from the counter hash
        const unsigned long long sd = seed + (seed_dev ? *seed_dev : 0ull);
        s += noise_mult * (uniform24((uint32_t)sd, (uint32_t)(sd >> 32), (uint32_t)r, 0xC817u) - 0.5f);
    }
    if (lane == 0) out[r] = s;
}

// gated-GELU of T5 v1.1's feed-forward (HuggingFace modeling_t5.py T5DenseGatedActDense: wo(gelu_new(wi_0 x) * wi_1 x)) on the output
// of ONE GEMM against the row-concatenated [wi_0 ; wi_1]: h (M, 2F) f32 -> out (M, F) T, out = gelu_new(h[:, :F]) * h[:, F:].
// gelu_new = 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) (transformers activations.py NewGELUActivation).  Not on the video hot
// path (the text encoder runs once per prompt): kept out of the GEMM epilogue on purpose.
template <typename TO>
__global__ __launch_bounds__(256) void gated_gelu_tanh_kernel(const float* __restrict__ h, int ldh, TO* __restrict__ out, int ldo, int M, int Fd) {
    const int fv = Fd >> 2;
    const long total = (long)M * fv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % fv) * 4;
        const long m = idx / fv;
        const f32x4 a = *reinterpret_cast<const f32x4*>(h + m * ldh + c), b = *reinterpret_cast<const f32x4*>(h + m * ldh + Fd + c);
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = a[r];
            y[r] = 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))) * b[r];
        }
        store4(out + m * ldo + c, y);
    }
}

}  // namespace pk
using namespace pk;

static inline int nblocks(long total) { long b = (total + 255) / 256; return (int)(b < 16384 ? (b > 0 ? b : 1) : 16384); }
#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int pk_peg(const float* x, const float* wt, const float* bias, float* out, void* out_t, int B, int T, int H, int W, int D,
                      int causal, void* stream) {
    if (!x || !wt || !bias || !out || B <= 0 || T <= 0 || H <= 0 || W <= 0 || D <= 0) return PK_EINVAL;
    if (D & 3) return PK_EALIGN;
    if (x == out) return PK_EINVAL;                      // stencil: not in-place
    const long total = (long)B * T * H * W * (D >> 2);
    const long rows = (long)B * T * H * (D >> 2);
    const dim3 rgrid(xcd_padded_grid((rows + 255) / 256));
    bf16* ot = reinterpret_cast<bf16*>(out_t);
    if (ot && (reinterpret_cast<uintptr_t>(ot) & 7)) return PK_EALIGN;
    if (W == 8) hipLaunchKernelGGL((peg_row_kernel<8>), rgrid, dim3(256), 0, STREAM(stream), x, wt, bias, out, ot, B, T, H, D, causal ? 2 : 1, rows);
    else if (W == 4) hipLaunchKernelGGL((peg_row_kernel<4>), rgrid, dim3(256), 0, STREAM(stream), x, wt, bias, out, ot, B, T, H, D, causal ? 2 : 1, rows);
    else if (W == 16) hipLaunchKernelGGL((peg_row_kernel<16>), rgrid, dim3(256), 0, STREAM(stream), x, wt, bias, out, ot, B, T, H, D, causal ? 2 : 1, rows);
    else hipLaunchKernelGGL(peg_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), x, wt, bias, out, ot, B, T, H, W, D, causal ? 2 : 1, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// the adjoint stencil on the row kernel (W in {4, 8, 16}); returns PK_EINVAL for other widths (the caller keeps its gather kernel)
extern "C" int pk_peg_adjoint(const float* dy, const float* wt, float* dx, int B, int T, int H, int W, int D, int causal, void* stream) {
    if (!dy || !wt || !dx || B <= 0 || T <= 0 || H <= 0 || D <= 0 || dy == dx) return PK_EINVAL;
    if (D & 3) return PK_EALIGN;
    const long rows = (long)B * T * H * (D >> 2);
    const dim3 rgrid(xcd_padded_grid((rows + 255) / 256));
    const int tf = 2 - (causal ? 2 : 1);
    if (W == 8) hipLaunchKernelGGL((peg_row_kernel<8>), rgrid, dim3(256), 0, STREAM(stream), dy, wt, nullptr, dx, nullptr, B, T, H, D, tf, rows, 1);
    else if (W == 4) hipLaunchKernelGGL((peg_row_kernel<4>), rgrid, dim3(256), 0, STREAM(stream), dy, wt, nullptr, dx, nullptr, B, T, H, D, tf, rows, 1);
    else if (W == 16) hipLaunchKernelGGL((peg_row_kernel<16>), rgrid, dim3(256), 0, STREAM(stream), dy, wt, nullptr, dx, nullptr, B, T, H, D, tf, rows, 1);
    else return PK_EINVAL;
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_lfq_encode(const float* x, int ldx, const float* wp, const float* bp, long long* ids, float* proj,
                             int M, int D, int cd, void* stream) {
    if (!x || !wp || !bp || !ids || M <= 0 || D <= 0) return PK_EINVAL;
    if ((D & 3) || (ldx & 3)) return PK_EALIGN;
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = STREAM(stream);
    switch (cd) {
#define PK_CASE(N) case N: hipLaunchKernelGGL((lfq_encode_kernel<N>), grid, block, 0, s, x, ldx, wp, bp, ids, proj, M, D); break;
        PK_CASE(4) PK_CASE(6) PK_CASE(8) PK_CASE(10) PK_CASE(12) PK_CASE(13) PK_CASE(14) PK_CASE(16) PK_CASE(18)
#undef PK_CASE
        default: return PK_EINVAL;
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_lfq_decode(const long long* ids, const float* wo, const float* bo, float* out, int M, int D, int cd,
                             const long long* ids_prime, int n_prime, int n, int pb, int pc, void* stream) {
    if (!ids || !wo || !bo || !out || M <= 0 || D <= 0 || cd <= 0 || cd > 62) return PK_EINVAL;
    if (ids_prime && (n_prime <= 0 || n <= 0 || M % (n_prime + n))) return PK_EINVAL;
    if (!ids_prime) { n_prime = 0; n = 0; }
    if ((pb > 0) != (pc > 0) || (pb > 0 && M % (pb * pc))) return PK_EINVAL;
    hipStream_t s = STREAM(stream);
    const int dv = D >> 2;
    const bool fast = (D & 3) == 0 && dv <= 256 && (256 % dv == 0 || dv == 256) && (cd == 8 || cd == 16) &&
                      (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(bo) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(wo) & 15) == 0;
    if (fast) {
        const int rpb = 256 / dv;                               // rows in flight per 256-thread block
        int blocks = (M + rpb - 1) / rpb;
        if (blocks > 512) blocks = 512;                         // 2 blocks per CU: the per-thread weight load is amortised over >= 4 rows at M = 4608
        if (cd == 16) hipLaunchKernelGGL((lfq_decode_kernel<16>), dim3(blocks), dim3(256), 0, s, ids_prime, n_prime, ids, n, wo, bo, out, M, D, blocks * rpb, pb, pc);
        else hipLaunchKernelGGL((lfq_decode_kernel<8>), dim3(blocks), dim3(256), 0, s, ids_prime, n_prime, ids, n, wo, bo, out, M, D, blocks * rpb, pb, pc);
    } else {
        hipLaunchKernelGGL(lfq_decode_generic_kernel, dim3(nblocks((long)M * D)), dim3(256), 0, s, ids_prime, n_prime, ids, n, wo, bo, out, M, D, cd, pb, pc);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_embed(const long long* ids_prime, int n_prime, const long long* ids, int n, int nb, const float* tok,
                        const float* pos, float* out, void* out_t, int S, int D, void* stream) {
    if (!ids || !tok || !pos || !out || S <= 0 || n <= 0 || nb <= 0 || D <= 0 || n_prime < 0 || (n_prime > 0 && !ids_prime)) return PK_EINVAL;
    if (D & 3) return PK_EALIGN;
    const long total = (long)S * (n_prime + n) * (D >> 2);
    hipLaunchKernelGGL(embed_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), ids_prime, n_prime, ids, n, nb, tok, pos, out, reinterpret_cast<bf16*>(out_t), D, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_cpb_input(const float* w0, const float* b0, float* out, int d0, int d1, int d2, int nd, int D, void* stream) {
    if (!w0 || !b0 || !out || d0 <= 0 || d1 <= 0 || d2 <= 0 || nd < 1 || nd > 3 || D <= 0) return PK_EINVAL;
    const long n = (long)d0 * d1 * d2;
    const long total = n * n * D;
    hipLaunchKernelGGL(cpb_input_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), w0, b0, out, d0, d1, d2, nd, D, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_critic_head(const float* x, int ldx, const float* w, const float* b, int D, int nb, int n_tot, int n_prime,
                              int has_null, float scale, const float* u, float noise_mult, unsigned long long seed,
                              const unsigned long long* seed_dev, float* out, void* stream) {
    if (!x || !w || !out || D <= 0 || nb <= 0 || n_tot <= n_prime || n_prime < 0) return PK_EINVAL;
    if ((D & 3) || (ldx & 3)) return PK_EALIGN;
    const int rows = nb * (n_tot - n_prime);
    hipLaunchKernelGGL(critic_head_kernel, dim3((rows + 3) / 4), dim3(256), 0, STREAM(stream), x, ldx, w, b, D, nb, n_tot, n_prime,
                       has_null, scale, u, noise_mult, seed, seed_dev, out);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_gated_gelu_tanh(const float* h, int ldh, void* out, int ldo, int out_kind, int M, int F, void* stream) {
    if (!h || !out || M <= 0 || F <= 0 || ldh < 2 * F || ldo < F) return PK_EINVAL;
    if ((F & 3) || (ldh & 3) || (ldo & 3) || (reinterpret_cast<uintptr_t>(h) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return PK_EALIGN;
    const long total = (long)M * (F >> 2);
    if (out_kind == 0) hipLaunchKernelGGL((gated_gelu_tanh_kernel<float>), dim3(nblocks(total)), dim3(256), 0, STREAM(stream), h, ldh, (float*)out, ldo, M, F);
    else hipLaunchKernelGGL((gated_gelu_tanh_kernel<bf16>), dim3(nblocks(total)), dim3(256), 0, STREAM(stream), h, ldh, (bf16*)out, ldo, M, F);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
