// Backward kernels of the MaskGit / TokenCritic trunk (the training step of phenaki_pytorch.py:562-687 under autograd;
// what each one differentiates is cited at the kernel).  Everything here except the operand packer is HBM / latency bound elementwise or
// row-reduction work; the matrix products of the backward pass are pk_gemm calls on operands laid down by pk_pack (host: train.py).
// Activations and gradients are f32 in HBM; reductions over rows are two-stage (per-block partials + pk_colsum) so that every gradient
// is bit-reproducible run to run -- the two exceptions (token-embedding and position-bias scatter) say so.
#include "common.hpp"

namespace pk {

#define STREAM(s) reinterpret_cast<hipStream_t>(s)
static inline int nblocks(long total) { long b = (total + 255) / 256; return (int)(b < 65535 ? (b > 0 ? b : 1) : 65535); }

// ---- pk_pack: f32 matrix -> GEMM operand image, optionally transposed, K zero-padded to the k-tile -------------------------------
// out[r][k] = (TR ? src[k][r] : src[r][k]) for k < K, 0 for K <= k < Kp; rows (optional) gathers SOURCE rows (src[rows[k]][r] / src[rows[r]][k]).   The dW product of a Linear, dW = dY^T X, takes BOTH its operands
// through this kernel (A = dY^T, "W" = X^T: the contraction index of pk_gemm is the contiguous one), dX = dY W takes W^T.
template <typename TO, bool TR>
__device__ __forceinline__ void pack_tile(float (*tile)[65], const float* __restrict__ src, long lds_, const int* __restrict__ rows, int R, int K, int Kp,
                                          TO* __restrict__ out, long ldo, int bx, int by) {
    const int r0 = by * 64, k0 = bx * 64, t = threadIdx.x;
    // 16-byte source loads when the rows allow it (round 6: four 4-byte loads per thread kept the transposes of a backward block at 1.9 TB/s)
    const bool vec = !(lds_ & 3) && !(reinterpret_cast<uintptr_t>(src) & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (TR) {
            const int kk = (t >> 4) + 16 * i, rr = (t & 15) * 4;
            const int k = k0 + kk, r = r0 + rr;
            const long sk = (k < K) ? (rows ? rows[k] : k) : 0;          // source row (optionally gathered)
            if (vec && k < K && r + 3 < R) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + sk * lds_ + r);
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[rr + j][kk] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[rr + j][kk] = (k < K && r + j < R) ? src[sk * lds_ + r + j] : 0.f;
            }
        } else {
            const int rr = (t >> 4) + 16 * i, kk = (t & 15) * 4;
            const int k = k0 + kk, r = r0 + rr;
            const long sr = (r < R) ? (rows ? rows[r] : r) : 0;
            if (vec && r < R && k + 3 < K) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + sr * lds_ + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[rr][kk + j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[rr][kk + j] = (r < R && k + j < K) ? src[sr * lds_ + k + j] : 0.f;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = (t >> 4) + 16 * i, kk = (t & 15) * 4;
        const int r = r0 + rr, k = k0 + kk;
        if (r < R && k < Kp) store4(out + (long)r * ldo + k, f32x4{tile[rr][kk], tile[rr][kk + 1], tile[rr][kk + 2], tile[rr][kk + 3]});
    }
}
template <typename TO, bool TR>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, long lds_, const int* __restrict__ rows, int R, int K, int Kp,
                                                   TO* __restrict__ out, long ldo) {
    __shared__ float tile[64][65];
    pack_tile<TO, TR>(tile, src, lds_, rows, R, K, Kp, out, ldo, blockIdx.x, blockIdx.y);
}

// ---- several pk_pack jobs in ONE launch (round 6: the training step issued ~440 single-matrix packs of 5-9 us each) -------------------------
// A job is one output block: out (R rows, columns [0, Kp)) = src or src^T, zero for K <= column < Kp; `tile0` = the launch-wide index of its first
// 64 x 64 tile, `tiles_x` = ceil(Kp / 64).  Two carriers of the job list: by value in the kernel arguments (<= PACK_JOBS jobs: the activation
// transposes of one backward block, pointers change every call) and a table in device memory (the persistent operand images of a whole
// Transformer's projection weights, written once per model and re-run every step: pk_pack_table).
struct PackJob {
    const float* src; void* out; long lds_, ldo; int R, K, Kp, flags /* bit 0 transpose, bits 1-2 kind */, tile0, tiles_x;
};
constexpr int PACK_JOBS = 8;
struct PackJobs { PackJob j[PACK_JOBS]; };

__device__ __forceinline__ void pack_job_tile(float (*tile)[65], const PackJob& d, int rel) {
    const int bx = rel % d.tiles_x, by = rel / d.tiles_x;
    const int kind = d.flags >> 1;
    const bool tr = d.flags & 1;
    if (kind == 0) {
        if (tr) pack_tile<float, true>(tile, d.src, d.lds_, nullptr, d.R, d.K, d.Kp, reinterpret_cast<float*>(d.out), d.ldo, bx, by);
        else pack_tile<float, false>(tile, d.src, d.lds_, nullptr, d.R, d.K, d.Kp, reinterpret_cast<float*>(d.out), d.ldo, bx, by);
    } else if (kind == 1) {
        if (tr) pack_tile<bf16, true>(tile, d.src, d.lds_, nullptr, d.R, d.K, d.Kp, reinterpret_cast<bf16*>(d.out), d.ldo, bx, by);
        else pack_tile<bf16, false>(tile, d.src, d.lds_, nullptr, d.R, d.K, d.Kp, reinterpret_cast<bf16*>(d.out), d.ldo, bx, by);
    } else {
        if (tr) pack_tile<bf16x3p, true>(tile, d.src, d.lds_, nullptr, d.R, d.K, d.Kp, reinterpret_cast<bf16x3p*>(d.out), d.ldo, bx, by);
        else pack_tile<bf16x3p, false>(tile, d.src, d.lds_, nullptr, d.R, d.K, d.Kp, reinterpret_cast<bf16x3p*>(d.out), d.ldo, bx, by);
    }
}
__global__ __launch_bounds__(256) void pack_jobs_kernel(const PackJobs a, int count) {
    __shared__ float tile[64][65];
    const int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int c = 1; c < PACK_JOBS; ++c)
        if (c < count && b >= a.j[c].tile0) i = c;
    pack_job_tile(tile, a.j[i], b - a.j[i].tile0);
}
__global__ __launch_bounds__(256) void pack_table_kernel(const PackJob* __restrict__ table, int count) {
    __shared__ float tile[64][65];
    const int b = blockIdx.x;
    int lo = 0, hi = count - 1;                                       // last job whose tile0 <= b (uniform over the block: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const PackJob d = table[lo];
    pack_job_tile(tile, d, b - d.tile0);
}

// ---- pk_colsum: out[c] (+)= scale * sum_r src[r][c], two deterministic stages -------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ src, long ld, int M, int N, float* __restrict__ part, int rpb) {
    __shared__ float red[4][64];
    const int t = threadIdx.x, c = blockIdx.x * 64 + (t & 63), rl = t >> 6;
    const int rb = blockIdx.y * rpb, re = min(M, rb + rpb);
    float a = 0.f;
    if (c < N)
        for (int r = rb + rl; r < re; r += 4) a += src[(long)r * ld + c];
    red[rl][t & 63] = a;
    __syncthreads();
    if (rl == 0 && c < N) part[(long)blockIdx.y * N + c] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}
// float4-addressable rows (round 6): 16 columns x 64 row lanes per block like colsum_one_kernel -- a 64-column gradient over 524 288 pixel rows (the
// discriminator's convolution biases) was 256 blocks of 64 active column lanes reading 4 bytes each: 64 us for 134 MB
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ src, long ld, int M, int N, float* __restrict__ part, int rpb) {
    __shared__ f32x4 red[64][4];
    const int t = threadIdx.x, cl = t & 3, rl = t >> 2, c = blockIdx.x * 16 + cl * 4;
    const int rb = blockIdx.y * rpb, re = min(M, rb + rpb);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < N)
        for (int r = rb + rl; r < re; r += 64) a += *reinterpret_cast<const f32x4*>(src + (long)r * ld + c);
    red[rl][cl] = a;
    __syncthreads();
    if (t < 16) {
        const int p = t >> 2;
        f32x4 s = red[16 * p][cl];
#pragma unroll
        for (int i = 1; i < 16; ++i) s += red[16 * p + i][cl];
        red[16 * p][cl] = s;
    }
    __syncthreads();
    if (t < 4 && c < N) *reinterpret_cast<f32x4*>(part + (long)blockIdx.y * N + c) = (red[0][cl] + red[16][cl]) + (red[32][cl] + red[48][cl]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int P, int N, float scale, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += part[(long)p * N + c];
    out[c] = accumulate ? out[c] + scale * s : scale * s;
}

// one-launch form for short inputs (round 6: the per-block partials of ln_bwd / peg_wgrad / attn_train_prep_bwd / bce_head are 144 ... 1024 rows --
// the two-stage form spent two 5-6 us launches on each of the step's 141 column sums).  A block owns 64 columns; 16 row lanes x 16 float4 column
// lanes (V4) or 4 row lanes x 64 columns; the lanes' sums are folded in lane order, so the result is as reproducible as the two-stage one.
// 16 columns per block (4 float4 lanes) x 64 row lanes: 32 blocks for a 512-column partial matrix (64 columns per block left 8 workgroups walking 36
// rows each: 12 us); the 64 lanes of a column are folded in a fixed order (16 lanes per thread, then 4 partial sums)
__device__ __forceinline__ void colsum16_block(f32x4 (*red)[4], const float* __restrict__ src, long ld, int M, int N, float scale, float* __restrict__ out,
                                               int accumulate, int bx) {
    const int t = threadIdx.x;
    const int cl = t & 3, rl = t >> 2, c = bx * 16 + cl * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c < N)
        for (int r = rl; r < M; r += 64) a += *reinterpret_cast<const f32x4*>(src + (long)r * ld + c);
    red[rl][cl] = a;
    __syncthreads();
    if (t < 16) {                                                      // thread (part p, column lane cl): row lanes 16 p .. 16 p + 15
        const int p = t >> 2;
        f32x4 s = red[16 * p][cl];
#pragma unroll
        for (int i = 1; i < 16; ++i) s += red[16 * p + i][cl];
        red[16 * p][cl] = s;                                           // (only this thread reads rows 16 p .. 16 p + 15 of column lane cl)
    }
    __syncthreads();
    if (t < 4 && c < N) {
        const f32x4 s = (red[0][cl] + red[16][cl]) + (red[32][cl] + red[48][cl]);
        f32x4* o = reinterpret_cast<f32x4*>(out + c);
        *o = accumulate ? *o + s * scale : s * scale;
    }
}
// several short column sums in ONE launch (the partial buffers a backward block's kernels leave behind: dgamma, dq_scale | dk_scale, PEG taps, ...)
struct ColsumJob { const float* src; float* out; long ld; int M, N, blk0; float scale; };
constexpr int COLSUM_JOBS = 8;
struct ColsumJobs { ColsumJob j[COLSUM_JOBS]; };
__global__ __launch_bounds__(256) void colsum_jobs_kernel(const ColsumJobs a, int count) {
    __shared__ f32x4 red[64][4];
    const int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int c = 1; c < COLSUM_JOBS; ++c)
        if (c < count && b >= a.j[c].blk0) i = c;
    const ColsumJob d = a.j[i];
    colsum16_block(red, d.src, d.ld, d.M, d.N, d.scale, d.out, 0, b - d.blk0);
}
template <bool V4>
__global__ __launch_bounds__(256) void colsum_one_kernel(const float* __restrict__ src, long ld, int M, int N, float scale, float* __restrict__ out, int accumulate) {
    __shared__ f32x4 red[64][4];
    const int t = threadIdx.x;
    if (V4) {
        colsum16_block(red, src, ld, M, N, scale, out, accumulate, blockIdx.x);
    } else {
        float* redf = reinterpret_cast<float*>(&red[0][0]);
        const int cl = t & 63, rl = t >> 6, c = blockIdx.x * 64 + cl;
        float a = 0.f;
        if (c < N)
            for (int r = rl; r < M; r += 4) a += src[(long)r * ld + c];
        redf[rl * 64 + cl] = a;
        __syncthreads();
        if (rl == 0 && c < N) {
            const float s = (redf[cl] + redf[64 + cl]) + (redf[128 + cl] + redf[192 + cl]);
            out[c] = accumulate ? out[c] + scale * s : scale * s;
        }
    }
}

// ---- LayerNorm backward (attention.py:29-36 gamma-only LayerNorm and nn.LayerNorm of the feed-forward, attention.py:47) ------------
// y = xhat * gamma + beta, xhat = (x - mean) * rstd:   g = dy * gamma,  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ add],
// dgamma = sum_rows dy * xhat, dbeta = sum_rows dy  (per-block partials; pk_colsum finishes them).  One wave per row, statistics recomputed.
template <int VMAX>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, long lddy, const float* __restrict__ add, long ldadd,
                                                     float* __restrict__ dx, long lddx, float* __restrict__ pg, float* __restrict__ pb,
                                                     float eps, int M, int D, int rpb) {
    __shared__ f32x4 red[4][VMAX * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nv = D >> 2;
    f32x4 ag[VMAX], ab[VMAX], gm[VMAX];
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        ag[i] = f32x4{0, 0, 0, 0};
        ab[i] = ag[i];
        const int c = lane + i * 64;
        gm[i] = c < nv ? *reinterpret_cast<const f32x4*>(gamma + c * 4) : ag[i];
    }
    const int rb = blockIdx.x * rpb, re = min(M, rb + rpb);
    for (int row = rb + wv; row < re; row += 4) {
        const float* xr = x + (long)row * ldx;
        const float* dr = dy + (long)row * lddy;
        f32x4 v[VMAX], g[VMAX];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VMAX; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4);
                g[i] = *reinterpret_cast<const f32x4*>(dr + c * 4);
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VMAX; ++i)
            if (lane + i * 64 < nv) {
                v[i] -= mean;
                q += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < VMAX; ++i)
            if (lane + i * 64 < nv) {
                v[i] *= rstd;                                        // xhat
                ab[i] += g[i];
                ag[i] += g[i] * v[i];
                g[i] *= gm[i];
                c1 += (g[i][0] + g[i][1]) + (g[i][2] + g[i][3]);
                c2 += (g[i][0] * v[i][0] + g[i][1] * v[i][1]) + (g[i][2] * v[i][2] + g[i][3] * v[i][3]);
            }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
#pragma unroll
        for (int i = 0; i < VMAX; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                f32x4 o = (g[i] - c1 - v[i] * c2) * rstd;
                if (add) o += *reinterpret_cast<const f32x4*>(add + (long)row * ldadd + c * 4);
                *reinterpret_cast<f32x4*>(dx + (long)row * lddx + c * 4) = o;
            }
        }
    }
    // cross-wave fold of the column sums, then one partial row per block
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !pb) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VMAX; ++i) red[wv][lane + i * 64] = pass == 0 ? ag[i] : ab[i];
        __syncthreads();
        float* dst = pass == 0 ? pg : pb;
        const long pst = (pb == pg + D) ? 2 * D : D;                      // pg | pb as the halves of one (parts, 2 D) buffer: one pk_colsum for both
        for (int c = threadIdx.x; c < nv; c += 256) {
            const f32x4 s4 = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            *reinterpret_cast<f32x4*>(dst + (long)blockIdx.x * pst + c * 4) = s4;
        }
    }
}

// ---- GEGLU (attention.py:40-43: x, gate = chunk(2); gelu(gate) * x) forward on stored pre-activations, and its backward ---------------
__device__ __forceinline__ float gelu_grad(float g) {
    return 0.5f * (1.0f + erff(g * 0.70710678118654752440f)) + g * 0.3989422804014327f * __expf(-0.5f * g * g);
}
__global__ __launch_bounds__(256) void geglu_kernel(const float* __restrict__ h, long ldh, int goff, float* __restrict__ out, long ldo, int M, int F4, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % F4) * 4;
        const long r = idx / F4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(h + r * ldh + c), gv = *reinterpret_cast<const f32x4*>(h + r * ldh + goff + c);
        *reinterpret_cast<f32x4*>(out + r * ldo + c) = f32x4{xv[0] * gelu_erf(gv[0]), xv[1] * gelu_erf(gv[1]), xv[2] * gelu_erf(gv[2]), xv[3] * gelu_erf(gv[3])};
    }
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const float* __restrict__ h, long ldh, int goff, const float* __restrict__ dout, long ldd,
                                                        float* __restrict__ dh, long lddh, int M, int F4, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % F4) * 4;
        const long r = idx / F4;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(h + r * ldh + c), gv = *reinterpret_cast<const f32x4*>(h + r * ldh + goff + c);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dout + r * ldd + c);
        f32x4 dxv, dgv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dxv[j] = dv[j] * gelu_erf(gv[j]);
            dgv[j] = dv[j] * xv[j] * gelu_grad(gv[j]);
        }
        *reinterpret_cast<f32x4*>(dh + r * lddh + c) = dxv;
        *reinterpret_cast<f32x4*>(dh + r * lddh + goff + c) = dgv;
    }
}

// ---- LeakyReLU backward from the activation's OUTPUT (the position-bias MLP, attention.py:243-247: sign(y) == sign(pre-activation)) ---
__global__ __launch_bounds__(256) void leaky_bwd_kernel(const float* __restrict__ y, long ldy, const float* __restrict__ dy, long lddy,
                                                        float* __restrict__ dz, long lddz, int N, float slope, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % N);
        const long r = idx / N;
        dz[r * lddz + c] = dy[r * lddy + c] * (y[r * ldy + c] > 0.f ? 1.0f : slope);
    }
}

// ---- C-ViViT reconstruction step (cvivit.py:585-591, LFQ straight-through): elementwise pieces -----------------------------------------
// out = (a - b) * scale [* *scale_dev]: the gradient of sum (a - b)^2 * (scale / 2) w.r.t. a
__global__ __launch_bounds__(256) void scaled_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, float scale,
                                                          const float* __restrict__ scale_dev, float* __restrict__ out, long total4) {
    const float sc = scale * (scale_dev ? *scale_dev : 1.0f);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + idx * 4), y = *reinterpret_cast<const f32x4*>(b + idx * 4);
        *reinterpret_cast<f32x4*>(out + idx * 4) = (x - y) * sc;
    }
}
// out = a * b (the summand of the patch LayerNorm's weight gradient: dz (.) x^)
__global__ __launch_bounds__(256) void mul_kernel(const float* a, const float* b, float* out, long total4) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256)
        *reinterpret_cast<f32x4*>(out + idx * 4) = *reinterpret_cast<const f32x4*>(a + idx * 4) * *reinterpret_cast<const f32x4*>(b + idx * 4);
}
// out = z > 0 ? +value : -value (the LFQ code of a projection)
__global__ __launch_bounds__(256) void sign_kernel(const float* __restrict__ z, float value, float* __restrict__ out, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) out[idx] = z[idx] > 0.f ? value : -value;
}

// ---- PEG backward (attention.py:57-85 + the residual of :323: y = x + dsconv(pad(x)) + b) ------------------------------------------------
// dx = dy + sum_taps w[tap] * dy[shifted the other way];  tfront = 2 causal / 1 centred frame padding, as in pk_peg
__global__ __launch_bounds__(256) void peg_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ wt, float* __restrict__ dx,
                                                      int B, int T, int H, int W, int D, int tfront, long total) {
    const int dv = D >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % dv) * 4;
        long p = idx / dv;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H); p /= H;
        const int t = (int)(p % T); const int b = (int)(p / T);
        f32x4 acc = *reinterpret_cast<const f32x4*>(dy + ((((long)b * T + t) * H + h) * W + w) * D + c);
        for (int dt = 0; dt < 3; ++dt) {
            const int ts = t - (dt - tfront);                       // the output position whose tap dt read input frame t
            if (ts < 0 || ts >= T) continue;
            for (int dh = 0; dh < 3; ++dh) {
                const int hs = h - (dh - 1);
                if (hs < 0 || hs >= H) continue;
                for (int dw = 0; dw < 3; ++dw) {
                    const int ws = w - (dw - 1);
                    if (ws < 0 || ws >= W) continue;
                    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((((long)b * T + ts) * H + hs) * W + ws) * D + c);
                    const f32x4 k = *reinterpret_cast<const f32x4*>(wt + (long)((dt * 3 + dh) * 3 + dw) * D + c);
                    acc += g * k;
                }
            }
        }
        *reinterpret_cast<f32x4*>(dx + idx * 4) = acc;
    }
}
// dW[tap][d] = sum_pos dy[pos][d] * x[pos + tap offset][d] as per-block partials part[block][27][D]
// grid.y = 3: block (b, dt) accumulates the 9 taps of temporal offset dt (round 6: 27 accumulators per thread = 108 VGPRs at one wave per SIMD and
// 288 blocks left the tap loads' latency exposed, 69 us for 64 M multiply-adds)
__global__ __launch_bounds__(256) void peg_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                        int B, int T, int H, int W, int D, int tfront, long rows, int rpb) {
    __shared__ f32x4 red[256];
    const int dv = D >> 2, nl = 256 / dv;
    const int cq = threadIdx.x % dv, rl = threadIdx.x / dv, c = cq * 4;
    const int dt = blockIdx.y;
    f32x4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f32x4{0, 0, 0, 0};
    const long rb = (long)blockIdx.x * rpb, re = rb + rpb < rows ? rb + rpb : rows;
    if (rl < nl)
        for (long pos = rb + rl; pos < re; pos += nl) {
            long p = pos;
            const int w = (int)(p % W); p /= W;
            const int h = (int)(p % H); p /= H;
            const int t = (int)(p % T); const int b = (int)(p / T);
            const int ts = t + dt - tfront;
            if (ts < 0 || ts >= T) continue;
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + pos * D + c);
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const int hs = h + dh - 1;
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const int ws = w + dw - 1;
                    if (hs >= 0 && hs < H && ws >= 0 && ws < W)
                        acc[dh * 3 + dw] += g * *reinterpret_cast<const f32x4*>(x + ((((long)b * T + ts) * H + hs) * W + ws) * D + c);
                }
            }
        }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        __syncthreads();
        red[threadIdx.x] = acc[k];
        __syncthreads();
        if (rl == 0) {
            f32x4 s = red[cq];
            for (int l = 1; l < nl; ++l) s += red[l * dv + cq];
            *reinterpret_cast<f32x4*>(part + ((long)blockIdx.x * 27 + dt * 9 + k) * D + c) = s;
        }
    }
}

// ---- token + position embedding backward (phenaki_pytorch.py:194-199: x = tok[ids] + pos[arange]; x = x * alpha + x.detach() * (1 - alpha))
// dpos[p] = alpha * sum_s dy[s, p] (deterministic);  dtok[id] += alpha * dy[row] by f32 atomics: the mask id collects thousands of rows, so
// the summation ORDER (not the set of terms) of this one gradient varies run to run
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dy, const long long* __restrict__ ids, float alpha,
                                                        float* __restrict__ dtok, float* __restrict__ dpos, int S, int n, int D, long total) {
    const int dv = D >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % dv) * 4;
        const int p = (int)(idx / dv);
        f32x4 a = f32x4{0, 0, 0, 0}, run = a;
        long long cur = -1;
        // consecutive sequences with the SAME id at this position (the mask id collects most rows) are summed before the atomics: the hot row sees
        // one add per run instead of one per sequence
        for (int s = 0; s < S; ++s) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + ((long)s * n + p) * D + c) * alpha;
            a += g;
            const long long id = ids[(long)s * n + p];
            if (id != cur) {
                if (cur >= 0) {
                    float* dst = dtok + cur * D + c;
                    atomicAdd(dst, run[0]); atomicAdd(dst + 1, run[1]); atomicAdd(dst + 2, run[2]); atomicAdd(dst + 3, run[3]);
                }
                cur = id;
                run = g;
            } else run += g;
        }
        if (cur >= 0) {
            float* dst = dtok + cur * D + c;
            atomicAdd(dst, run[0]); atomicAdd(dst + 1, run[1]); atomicAdd(dst + 2, run[2]); atomicAdd(dst + 3, run[3]);
        }
        *reinterpret_cast<f32x4*>(dpos + (long)p * D + c) = a;
    }
}

// ---- continuous position bias: gather of the relative-position table into (heads, n, n) and its adjoint (attention.py:257-275) ---------
// bias[h][i][j] = tab[code[i] - code[j] + off][h]  (tab rows = relative positions, the MLP's output layout)
__global__ __launch_bounds__(256) void bias_gather_kernel(const float* __restrict__ tab, int ldt, const int* __restrict__ code, int off, float* __restrict__ out,
                                                          int heads, int n, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx % n);
        const int i = (int)((idx / n) % n);
        const int h = (int)(idx / ((long)n * n));
        out[idx] = tab[(long)(code[i] - code[j] + off) * ldt + h];
    }
}
__global__ __launch_bounds__(256) void bias_scatter_kernel(const float* __restrict__ dbias, const int* __restrict__ code, int off, float* __restrict__ dtab, int ldt,
                                                           int heads, int n, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int j = (int)(idx % n);
        const int i = (int)((idx / n) % n);
        const int h = (int)(idx / ((long)n * n));
        atomicAdd(dtab + (long)(code[i] - code[j] + off) * ldt + h, dbias[idx]);
    }
}

// ---- dst[rows[m]] = src[m] (rows of the masked positions back into the full (b n, D) gradient; dst zeroed by the caller) ----------------------
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, long lds_, const int* __restrict__ rows, float* __restrict__ dst, long ldd,
                                                           int D4, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % D4) * 4;
        const long m = idx / D4;
        *reinterpret_cast<f32x4*>(dst + (long)rows[m] * ldd + c) = *reinterpret_cast<const f32x4*>(src + m * lds_ + c);
    }
}

// ---- out[e] = sum_s src[s][e] (the position bias is shared by every sequence of the batch: dbias = sum_s dS_s) -----------------------------
__global__ __launch_bounds__(256) void sum_batch_kernel(const float* __restrict__ src, long stride, int S, float* __restrict__ out, long E4) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < E4; idx += (long)gridDim.x * 256) {
        f32x4 a = *reinterpret_cast<const f32x4*>(src + idx * 4);
        for (int s = 1; s < S; ++s) a += *reinterpret_cast<const f32x4*>(src + (long)s * stride + idx * 4);
        *reinterpret_cast<f32x4*>(out + idx * 4) = a;
    }
}

// several pk_sum_batch jobs in ONE launch (the K-slice partials of the weight gradients of one backward block): blk0 = first block of the job
struct SumJob { const float* src; float* out; long stride, E4; int S, blk0; };
constexpr int SUM_JOBS = 8;
struct SumJobs { SumJob j[SUM_JOBS]; };
__global__ __launch_bounds__(256) void sum_batch_jobs_kernel(const SumJobs a, int count) {
    const int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int c = 1; c < SUM_JOBS; ++c)
        if (c < count && b >= a.j[c].blk0) i = c;
    const SumJob d = a.j[i];
    const long idx = (long)(b - d.blk0) * 256 + threadIdx.x;
    if (idx >= d.E4) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(d.src + idx * 4);
    for (int k = 1; k < d.S; ++k) s += *reinterpret_cast<const f32x4*>(d.src + (long)k * d.stride + idx * 4);
    *reinterpret_cast<f32x4*>(d.out + idx * 4) = s;
}

// the K-slice sums AND the short column sums of a backward block in ONE launch (round 6): blocks [0, nsum) run the sum jobs, the rest the column sums
__global__ __launch_bounds__(256) void reduce_jobs_kernel(const SumJobs a, int nsum_jobs, int nsum_blocks, const ColsumJobs c, int ncol_jobs) {
    __shared__ f32x4 red[64][4];
    const int b = blockIdx.x;
    if (b < nsum_blocks) {
        int i = 0;
#pragma unroll
        for (int k = 1; k < SUM_JOBS; ++k)
            if (k < nsum_jobs && b >= a.j[k].blk0) i = k;
        const SumJob d = a.j[i];
        const long idx = (long)(b - d.blk0) * 256 + threadIdx.x;
        if (idx >= d.E4) return;
        f32x4 s = *reinterpret_cast<const f32x4*>(d.src + idx * 4);
        for (int k = 1; k < d.S; ++k) s += *reinterpret_cast<const f32x4*>(d.src + (long)k * d.stride + idx * 4);
        *reinterpret_cast<f32x4*>(d.out + idx * 4) = s;
        return;
    }
    const int bc = b - nsum_blocks;
    int i = 0;
#pragma unroll
    for (int k = 1; k < COLSUM_JOBS; ++k)
        if (k < ncol_jobs && bc >= c.j[k].blk0) i = k;
    const ColsumJob d = c.j[i];
    colsum16_block(red, d.src, d.ld, d.M, d.N, d.scale, d.out, 0, bc - d.blk0);
}

// ---- critic head + BCE-with-logits, forward and backward in one pass (phenaki_pytorch.py:246-249 / :306-336 to_pred, :673-676) --------------
// logit = e . w + b;  loss_row = max(z, 0) - z y + log(1 + exp(-|z|));  dz = (sigmoid(z) - y) * scale;  de = dz * w;  dw / db as block partials
__global__ __launch_bounds__(256) void bce_head_kernel(const float* __restrict__ e, long lde, const float* __restrict__ w, const float* __restrict__ b,
                                                       const float* __restrict__ labels, float scale_, const float* __restrict__ scale_dev, float* __restrict__ logits, float* __restrict__ loss_rows,
                                                       float* __restrict__ de, long ldde, float* __restrict__ pw, float* __restrict__ pb, int M, int D, int rpb) {
    __shared__ f32x4 red[4][256];
    __shared__ float redb[4];
    const float scale = scale_dev ? scale_ * scale_dev[0] : scale_;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nv = D >> 2;
    f32x4 aw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) aw[i] = f32x4{0, 0, 0, 0};
    float abias = 0.f;
    const int rb = blockIdx.x * rpb, re = min(M, rb + rpb);
    for (int row = rb + wv; row < re; row += 4) {
        f32x4 v[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                v[i] = *reinterpret_cast<const f32x4*>(e + (long)row * lde + c * 4);
                const f32x4 wv4 = *reinterpret_cast<const f32x4*>(w + c * 4);
                s += (v[i][0] * wv4[0] + v[i][1] * wv4[1]) + (v[i][2] * wv4[2] + v[i][3] * wv4[3]);
            }
        }
        const float z = wave_sum(s) + (b ? b[0] : 0.f);
        const float y = labels ? labels[row] : 0.f;
        const float dz = labels ? (1.0f / (1.0f + __expf(-z)) - y) * scale : 0.f;
        if (lane == 0) {
            if (logits) logits[row] = z;
            if (loss_rows) loss_rows[row] = fmaxf(z, 0.f) - z * y + log1pf(__expf(-fabsf(z)));
        }
        abias += dz;
        if (de) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = lane + i * 64;
                if (c < nv) {
                    aw[i] += v[i] * dz;
                    *reinterpret_cast<f32x4*>(de + (long)row * ldde + c * 4) = *reinterpret_cast<const f32x4*>(w + c * 4) * dz;
                }
            }
        }
    }
    if (!pw) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wv][lane + i * 64] = aw[i];
    if (lane == 0) redb[wv] = abias;
    __syncthreads();
    for (int c = threadIdx.x; c < nv; c += 256)
        *reinterpret_cast<f32x4*>(pw + (long)blockIdx.x * D + c * 4) = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (threadIdx.x == 0) pb[blockIdx.x] = (redb[0] + redb[1]) + (redb[2] + redb[3]);
}

// ---- AdamW / Adam parameter update (reference optimizer.py:11-37 -> torch.optim.AdamW / Adam; one launch per parameter tensor) ----------------
// m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p = p (1 - lr wd) - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (round 6: a 16-bytes-per-lane form of this kernel measured 10.7 vs 10.2 us per launch -- the 4-byte form already streams the step's 4.3 GB at the rate the
// part sustains for 7 interleaved streams; removed)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    float lr, float b1, float b2, float eps, float wd, float bc1, float rsqrt_bc2, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        p[i] = p[i] * (1.0f - lr * wd) - (lr / bc1) * (mi / denom);
    }
}

// many small tensors in ONE launch: the table travels by value in the kernel arguments (no device-side pointer table to upload); block b
// updates chunk bc[b] (ADAM_CHUNK elements) of tensor bt[b]
constexpr int ADAM_T = 40, ADAM_B = 512, ADAM_CHUNK = 2048;
struct AdamMulti {
    float* p[ADAM_T]; const float* g[ADAM_T]; float* m[ADAM_T]; float* v[ADAM_T];
    int n[ADAM_T];
    unsigned short bc[ADAM_B];
    unsigned char bt[ADAM_B];
};
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamMulti a, float lr, float b1, float b2, float eps, float wd, float bc1, float rsqrt_bc2) {
    const int t = a.bt[blockIdx.x];
    const long base = (long)a.bc[blockIdx.x] * ADAM_CHUNK;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const long n = a.n[t];
#pragma unroll
    for (int u = 0; u < ADAM_CHUNK / 256; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i >= n) break;
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        p[i] = p[i] * (1.0f - lr * wd) - (lr / bc1) * (mi / denom);
    }
}

// the LARGE tensors of a step (>= 256 Ki elements: every projection weight, the embedding tables) several per launch (round 6): the block -> tensor
// map is a prefix array of block counts carried by value (no per-block table, so the grid is not limited by the kernel-argument size as in
// adamw_multi_kernel: a block still owns only 1024 elements, 4 per thread, and the launch streams like the one-tensor kernel)
constexpr int ADAM_BIG_T = 32, ADAM_BIG_CHUNK = 1024;
struct AdamBig {
    float* p[ADAM_BIG_T]; const float* g[ADAM_BIG_T]; float* m[ADAM_BIG_T]; float* v[ADAM_BIG_T];
    long n[ADAM_BIG_T];
    int blk0[ADAM_BIG_T];
};
__global__ __launch_bounds__(256) void adamw_big_kernel(const AdamBig a, int count, float lr, float b1, float b2, float eps, float wd, float bc1, float rsqrt_bc2) {
    const int b = blockIdx.x;
    int t = 0;
#pragma unroll
    for (int c = 1; c < ADAM_BIG_T; ++c)
        if (c < count && b >= a.blk0[c]) t = c;
    float* __restrict__ p = a.p[t];
    const float* __restrict__ g = a.g[t];
    float* __restrict__ m = a.m[t];
    float* __restrict__ v = a.v[t];
    const long n = a.n[t], base = (long)(b - a.blk0[t]) * ADAM_BIG_CHUNK;
#pragma unroll
    for (int u = 0; u < ADAM_BIG_CHUNK / 256; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i >= n) break;
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        p[i] = p[i] * (1.0f - lr * wd) - (lr / bc1) * (mi / denom);
    }
}

}  // namespace pk

using namespace pk;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// kind: 0 f32, 1 bf16, 2 split-bf16 image (bf16x3p).  out has R rows of ldo elements; columns [0, K) are filled from src, [K, Kp) zeroed
extern "C" int pk_pack(const float* src, long lds_, const int* rows, int R, int K, int transpose, void* out, long ldo, int Kp, int kind, void* stream) {
    if (!src || !out || R <= 0 || K <= 0 || Kp < K || ldo < Kp || kind < 0 || kind > 2) return PK_EINVAL;
    if ((Kp & 3) || (ldo & 3) || (reinterpret_cast<uintptr_t>(out) & (kind == 2 ? 127 : 15)) || (kind == 2 && ((ldo & 31) || (Kp & 31)))) return PK_EALIGN;
    const dim3 grid((Kp + 63) / 64, (R + 63) / 64);
    hipStream_t s = STREAM(stream);
#define PK_PACK(TO, TR) hipLaunchKernelGGL((pack_kernel<TO, TR>), grid, dim3(256), 0, s, src, lds_, rows, R, K, Kp, reinterpret_cast<TO*>(out), ldo)
    if (kind == 0) { if (transpose) PK_PACK(float, true); else PK_PACK(float, false); }
    else if (kind == 1) { if (transpose) PK_PACK(bf16, true); else PK_PACK(bf16, false); }
    else { if (transpose) PK_PACK(bf16x3p, true); else PK_PACK(bf16x3p, false); }
#undef PK_PACK
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// `count` pack jobs in one launch.  jobs: HOST array of `count` PkPackJob (include/phenaki_hip.h; tile0 / tiles_x are filled in here).
// count <= 8: the jobs travel in the kernel arguments.
static int pack_job_check(const PackJob& d) {
    const int kind = d.flags >> 1;
    if (!d.src || !d.out || d.R <= 0 || d.K <= 0 || d.Kp < d.K || d.ldo < d.Kp || kind < 0 || kind > 2) return PK_EINVAL;
    if ((d.Kp & 3) || (d.ldo & 3) || (reinterpret_cast<uintptr_t>(d.out) & 15) || (kind == 2 && (d.ldo & 31))) return PK_EALIGN;
    return PK_OK;
}
extern "C" int pk_pack_multi(const void* jobs, int count, void* stream) {
    if (!jobs || count <= 0 || count > PACK_JOBS) return PK_EINVAL;
    PackJobs a;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        a.j[i] = reinterpret_cast<const PackJob*>(jobs)[i];
        const int rc = pack_job_check(a.j[i]);
        if (rc != PK_OK) return rc;
        a.j[i].tiles_x = (a.j[i].Kp + 63) / 64;
        a.j[i].tile0 = tiles;
        tiles += a.j[i].tiles_x * ((a.j[i].R + 63) / 64);
    }
    for (int i = count; i < PACK_JOBS; ++i) a.j[i] = a.j[0];
    hipLaunchKernelGGL(pack_jobs_kernel, dim3(tiles), dim3(256), 0, STREAM(stream), a, count);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
// fill tile0 / tiles_x of a HOST job array (any count) and return the launch's tile count (< 0: error code) -- the caller copies the array to
// device memory once and replays it with pk_pack_table every step
extern "C" int pk_pack_table_prepare(void* jobs, int count) {
    if (!jobs || count <= 0) return PK_EINVAL;
    PackJob* j = reinterpret_cast<PackJob*>(jobs);
    long tiles = 0;
    for (int i = 0; i < count; ++i) {
        const int rc = pack_job_check(j[i]);
        if (rc != PK_OK) return rc;
        j[i].tiles_x = (j[i].Kp + 63) / 64;
        j[i].tile0 = (int)tiles;
        tiles += (long)j[i].tiles_x * ((j[i].R + 63) / 64);
        if (tiles > 0x7fffffffL) return PK_EINVAL;
    }
    return (int)tiles;
}
extern "C" int pk_pack_table(const void* dev_table, int count, int tiles, void* stream) {
    if (!dev_table || count <= 0 || tiles <= 0) return PK_EINVAL;
    hipLaunchKernelGGL(pack_table_kernel, dim3((unsigned)tiles), dim3(256), 0, STREAM(stream), reinterpret_cast<const PackJob*>(dev_table), count);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// out[c] (+)= scale * sum_r src[r][c];  work: >= P * N floats, P = pk_colsum_parts(M)
extern "C" int pk_colsum_parts(int M) { int p = (M + 63) / 64; return p < 1 ? 1 : (p > 256 ? 256 : p); }
extern "C" int pk_colsum(const float* src, long ld, int M, int N, float scale, float* out, int accumulate, float* work, void* stream) {
    if (!src || !out || !work || M <= 0 || N <= 0) return PK_EINVAL;
    hipStream_t s = STREAM(stream);
    const bool v4 = !(N & 3) && !(ld & 3) && al16(src) && al16(out);
    if (M <= (v4 ? 8192 : 256)) {                                        // short input (block partials of another kernel, a batch's rows): one launch
        if (v4) hipLaunchKernelGGL((colsum_one_kernel<true>), dim3((N + 15) / 16), dim3(256), 0, s, src, ld, M, N, scale, out, accumulate);
        else hipLaunchKernelGGL((colsum_one_kernel<false>), dim3((N + 63) / 64), dim3(256), 0, s, src, ld, M, N, scale, out, accumulate);
        PK_CHECK_LAUNCH();
        return PK_OK;
    }
    const int P = pk_colsum_parts(M), rpb = (M + P - 1) / P;
    if (v4 && al16(work)) {                                              // both stages on 16 x 64 blocks (the second one = the one-launch kernel over the partials)
        hipLaunchKernelGGL(colsum_partial4_kernel, dim3((N + 15) / 16, P), dim3(256), 0, s, src, ld, M, N, work, rpb);
        hipLaunchKernelGGL((colsum_one_kernel<true>), dim3((N + 15) / 16), dim3(256), 0, s, work, (long)N, P, N, scale, out, accumulate);
        PK_CHECK_LAUNCH();
        return PK_OK;
    }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, P), dim3(256), 0, s, src, ld, M, N, work, rpb);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, s, work, P, N, scale, out, accumulate);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// dx = [add +] LayerNorm backward; pg / pb (pb may be null: the gamma-only LayerNorm's beta is a buffer): (pk_ln_bwd_parts(M), D) partials
// 8 rows per block up to 1024 blocks (round 6: 32 rows per block = 144 blocks of 4 waves at M = 4608, half the CUs idle and 8 dependent rows per wave:
// 25.8 us for 38 MB; the partials stay within pk_colsum's one-launch bound)
extern "C" int pk_ln_bwd_parts(int M) { int p = (M + 7) / 8; return p < 1 ? 1 : (p > 1024 ? 1024 : p); }
extern "C" int pk_layernorm_bwd(const float* x, long ldx, const float* gamma, const float* dy, long lddy, const float* add, long ldadd,
                                float* dx, long lddx, float* pg, float* pb, float eps, int M, int D, void* stream) {
    if (!x || !gamma || !dy || !dx || !pg || M <= 0 || D <= 0 || D > 1024) return PK_EINVAL;
    if ((D & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3) || (ldadd & 3) || !al16(x) || !al16(dy) || !al16(dx) || !al16(gamma) || (add && !al16(add)) || !al16(pg) || (pb && !al16(pb))) return PK_EALIGN;
    const int P = pk_ln_bwd_parts(M), rpb = (M + P - 1) / P;
    hipStream_t s = STREAM(stream);
    if (D <= 512) hipLaunchKernelGGL((ln_bwd_kernel<2>), dim3(P), dim3(256), 0, s, x, ldx, gamma, dy, lddy, add, ldadd, dx, lddx, pg, pb, eps, M, D, rpb);
    else hipLaunchKernelGGL((ln_bwd_kernel<4>), dim3(P), dim3(256), 0, s, x, ldx, gamma, dy, lddy, add, ldadd, dx, lddx, pg, pb, eps, M, D, rpb);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// h (M, >= goff + F) f32: value columns [0, F), gate columns [goff, goff + F);  F % 4 == 0
extern "C" int pk_geglu(const float* h, long ldh, int goff, float* out, long ldo, int M, int F, void* stream) {
    if (!h || !out || M <= 0 || F <= 0 || goff < F) return PK_EINVAL;
    if ((F & 3) || (goff & 3) || (ldh & 3) || (ldo & 3) || !al16(h) || !al16(out)) return PK_EALIGN;
    const long total = (long)M * (F >> 2);
    hipLaunchKernelGGL(geglu_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), h, ldh, goff, out, ldo, M, F >> 2, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
extern "C" int pk_geglu_bwd(const float* h, long ldh, int goff, const float* dout, long ldd, float* dh, long lddh, int M, int F, void* stream) {
    if (!h || !dout || !dh || M <= 0 || F <= 0 || goff < F) return PK_EINVAL;
    if ((F & 3) || (goff & 3) || (ldh & 3) || (ldd & 3) || (lddh & 3) || !al16(h) || !al16(dout) || !al16(dh)) return PK_EALIGN;
    const long total = (long)M * (F >> 2);
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), h, ldh, goff, dout, ldd, dh, lddh, M, F >> 2, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_leaky_bwd(const float* y, long ldy, const float* dy, long lddy, float* dz, long lddz, int M, int N, float slope, void* stream) {
    if (!y || !dy || !dz || M <= 0 || N <= 0) return PK_EINVAL;
    const long total = (long)M * N;
    hipLaunchKernelGGL(leaky_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), y, ldy, dy, lddy, dz, lddz, N, slope, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_scaled_diff(const float* a, const float* b, float scale, const float* scale_dev, float* out, long n, void* stream) {
    if (!a || !b || !out || n <= 0) return PK_EINVAL;
    if ((n & 3) || !al16(a) || !al16(b) || !al16(out)) return PK_EALIGN;
    hipLaunchKernelGGL(scaled_diff_kernel, dim3(nblocks(n >> 2)), dim3(256), 0, STREAM(stream), a, b, scale, scale_dev, out, n >> 2);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_mul(const float* a, const float* b, float* out, long n, void* stream) {
    if (!a || !b || !out || n <= 0) return PK_EINVAL;
    if ((n & 3) || !al16(a) || !al16(b) || !al16(out)) return PK_EALIGN;
    hipLaunchKernelGGL(mul_kernel, dim3(nblocks(n >> 2)), dim3(256), 0, STREAM(stream), a, b, out, n >> 2);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_sign(const float* z, float value, float* out, long n, void* stream) {
    if (!z || !out || n <= 0) return PK_EINVAL;
    hipLaunchKernelGGL(sign_kernel, dim3(nblocks(n)), dim3(256), 0, STREAM(stream), z, value, out, n);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_peg_adjoint(const float* dy, const float* wt, float* dx, int B, int T, int H, int W, int D, int causal, void* stream);
// dx = dy + transposed stencil of dy;  part: (pk_peg_wgrad_parts(rows), 27, D) partial tap gradients (finish with pk_colsum over 27 * D columns)
// ~16 positions per block (2 row lanes at D = 512): 288 blocks at 4608 positions -- 36 blocks of 128 positions left 7 of 8 CUs idle (346 us)
extern "C" int pk_peg_wgrad_parts(long rows) { long p = (rows + 15) / 16; return (int)(p < 1 ? 1 : (p > 1024 ? 1024 : p)); }
extern "C" int pk_peg_bwd(const float* dy, const float* x, const float* wt, float* dx, float* part, int B, int T, int H, int W, int D, int causal, void* stream) {
    if (!dy || !wt || !dx || B <= 0 || T <= 0 || H <= 0 || W <= 0 || D <= 0 || (part && !x)) return PK_EINVAL;
    const int dv = D >> 2;
    if ((D & 3) || !al16(dy) || !al16(wt) || !al16(dx) || (x && !al16(x)) || (part && !al16(part))) return PK_EALIGN;
    if (part && (dv > 256 || 256 % dv)) return PK_EINVAL;
    hipStream_t s = STREAM(stream);
    const long rows = (long)B * T * H * W, total = rows * dv;
    // dx: the adjoint stencil on the forward's row kernel where it exists (W in {4, 8, 16}), else the 27-gather kernel
    if (pk_peg_adjoint(dy, wt, dx, B, T, H, W, D, causal, stream) != PK_OK)
        hipLaunchKernelGGL(peg_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, s, dy, wt, dx, B, T, H, W, D, causal ? 2 : 1, total);
    if (part) {
        const int P = pk_peg_wgrad_parts(rows);
        const int rpb = (int)((rows + P - 1) / P);
        hipLaunchKernelGGL(peg_wgrad_kernel, dim3(P, 3), dim3(256), 0, s, dy, x, part, B, T, H, W, D, causal ? 2 : 1, rows, rpb);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// dtok must be zeroed by the caller (rows the batch does not touch stay zero); dpos (n, D) is overwritten
extern "C" int pk_embed_bwd(const float* dy, const long long* ids, float alpha, float* dtok, float* dpos, int S, int n, int D, void* stream) {
    if (!dy || !ids || !dtok || !dpos || S <= 0 || n <= 0 || D <= 0) return PK_EINVAL;
    if ((D & 3) || !al16(dy) || !al16(dpos)) return PK_EALIGN;
    const long total = (long)n * (D >> 2);
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), dy, ids, alpha, dtok, dpos, S, n, D, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_bias_gather(const float* tab, int ldt, const int* code, int off, float* out, int heads, int n, void* stream) {
    if (!tab || !code || !out || heads <= 0 || n <= 0 || ldt < heads) return PK_EINVAL;
    const long total = (long)heads * n * n;
    hipLaunchKernelGGL(bias_gather_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), tab, ldt, code, off, out, heads, n, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
// dtab must be zeroed by the caller; f32 atomics (up to n collisions per entry: summation order varies run to run)
extern "C" int pk_bias_scatter(const float* dbias, const int* code, int off, float* dtab, int ldt, int heads, int n, void* stream) {
    if (!dbias || !code || !dtab || heads <= 0 || n <= 0 || ldt < heads) return PK_EINVAL;
    const long total = (long)heads * n * n;
    hipLaunchKernelGGL(bias_scatter_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), dbias, code, off, dtab, ldt, heads, n, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_scatter_rows(const float* src, long lds_, const int* rows, float* dst, long ldd, int M, int D, void* stream) {
    if (!src || !rows || !dst || M <= 0 || D <= 0) return PK_EINVAL;
    if ((D & 3) || (lds_ & 3) || (ldd & 3) || !al16(src) || !al16(dst)) return PK_EALIGN;
    const long total = (long)M * (D >> 2);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(nblocks(total)), dim3(256), 0, STREAM(stream), src, lds_, rows, dst, ldd, D >> 2, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_sum_batch(const float* src, long stride, int S, float* out, long E, void* stream) {
    if (!src || !out || S <= 0 || E <= 0) return PK_EINVAL;
    if ((E & 3) || (stride & 3) || !al16(src) || !al16(out)) return PK_EALIGN;
    hipLaunchKernelGGL(sum_batch_kernel, dim3(nblocks(E >> 2)), dim3(256), 0, STREAM(stream), src, stride, S, out, E >> 2);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// `count` (<= 8) short column sums in one launch: out[c] = scale * sum_{r < M} src[r][c], M <= 8192, N % 4 == 0, float4-addressable rows (the
// one-launch form of pk_colsum).  jobs: HOST array of PkColsumJob (blk0 is filled in here)
extern "C" int pk_colsum_multi(const void* jobs, int count, void* stream) {
    if (!jobs || count <= 0 || count > COLSUM_JOBS) return PK_EINVAL;
    ColsumJobs a;
    long blocks = 0;
    for (int i = 0; i < count; ++i) {
        a.j[i] = reinterpret_cast<const ColsumJob*>(jobs)[i];
        ColsumJob& d = a.j[i];
        if (!d.src || !d.out || d.M <= 0 || d.M > 8192 || d.N <= 0) return PK_EINVAL;
        if ((d.N & 3) || (d.ld & 3) || !al16(d.src) || !al16(d.out)) return PK_EALIGN;
        d.blk0 = (int)blocks;
        blocks += (d.N + 15) / 16;
    }
    for (int i = count; i < COLSUM_JOBS; ++i) a.j[i] = a.j[0];
    hipLaunchKernelGGL(colsum_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM(stream), a, count);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// both job lists of a backward block (<= 8 each; either may be empty) in one launch: pk_sum_batch_multi + pk_colsum_multi
extern "C" int pk_reduce_multi(const void* sum_jobs, int nsum, const void* col_jobs, int ncol, void* stream) {
    if (nsum < 0 || ncol < 0 || nsum > SUM_JOBS || ncol > COLSUM_JOBS || (nsum > 0 && !sum_jobs) || (ncol > 0 && !col_jobs) || nsum + ncol == 0) return PK_EINVAL;
    SumJobs a;
    ColsumJobs c;
    long sb = 0, cb = 0;
    for (int i = 0; i < nsum; ++i) {
        a.j[i] = reinterpret_cast<const SumJob*>(sum_jobs)[i];
        SumJob& d = a.j[i];
        if (!d.src || !d.out || d.S <= 0 || d.E4 <= 0) return PK_EINVAL;
        if ((d.stride & 3) || !al16(d.src) || !al16(d.out)) return PK_EALIGN;
        d.blk0 = (int)sb;
        sb += (d.E4 + 255) / 256;
    }
    for (int i = 0; i < ncol; ++i) {
        c.j[i] = reinterpret_cast<const ColsumJob*>(col_jobs)[i];
        ColsumJob& d = c.j[i];
        if (!d.src || !d.out || d.M <= 0 || d.M > 8192 || d.N <= 0) return PK_EINVAL;
        if ((d.N & 3) || (d.ld & 3) || !al16(d.src) || !al16(d.out)) return PK_EALIGN;
        d.blk0 = (int)cb;
        cb += (d.N + 15) / 16;
    }
    if (sb + cb > 0x7fffffffL) return PK_EINVAL;
    for (int i = nsum; i < SUM_JOBS; ++i) a.j[i] = a.j[nsum > 0 ? 0 : i];
    for (int i = ncol; i < COLSUM_JOBS; ++i) c.j[i] = c.j[ncol > 0 ? 0 : i];
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3((unsigned)(sb + cb)), dim3(256), 0, STREAM(stream), a, nsum, (int)sb, c, ncol);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// `count` (<= 8) pk_sum_batch jobs in one launch.  jobs: HOST array of PkSumJob (blk0 is filled in here)
extern "C" int pk_sum_batch_multi(const void* jobs, int count, void* stream) {
    if (!jobs || count <= 0 || count > SUM_JOBS) return PK_EINVAL;
    SumJobs a;
    long blocks = 0;
    for (int i = 0; i < count; ++i) {
        a.j[i] = reinterpret_cast<const SumJob*>(jobs)[i];
        SumJob& d = a.j[i];
        if (!d.src || !d.out || d.S <= 0 || d.E4 <= 0) return PK_EINVAL;
        if ((d.stride & 3) || !al16(d.src) || !al16(d.out)) return PK_EALIGN;
        d.blk0 = (int)blocks;
        blocks += (d.E4 + 255) / 256;
    }
    if (blocks > 0x7fffffffL) return PK_EINVAL;
    for (int i = count; i < SUM_JOBS; ++i) a.j[i] = a.j[0];
    hipLaunchKernelGGL(sum_batch_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM(stream), a, count);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// labels null: logits only (inference-style forward).  pw (pk_ln_bwd_parts(M), D) / pb (pk_ln_bwd_parts(M),) partials of dw / db
extern "C" int pk_bce_head(const float* e, long lde, const float* w, const float* b, const float* labels, float scale, const float* scale_dev, float* logits,
                           float* loss_rows,
                           float* de, long ldde, float* pw, float* pb, int M, int D, void* stream) {
    if (!e || !w || M <= 0 || D <= 0 || D > 1024 || (de && !labels) || (pw && (!pb || !de))) return PK_EINVAL;
    if ((D & 3) || (lde & 3) || (ldde & 3) || !al16(e) || !al16(w) || (de && !al16(de)) || (pw && !al16(pw))) return PK_EALIGN;
    const int P = pk_ln_bwd_parts(M), rpb = (M + P - 1) / P;
    hipLaunchKernelGGL(bce_head_kernel, dim3(P), dim3(256), 0, STREAM(stream), e, lde, w, b, labels, scale, scale_dev, logits, loss_rows, de, ldde, pw, pb, M, D, rpb);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// the same update for `count` tensors of one hyper-parameter set and step: table[i] = {p, g, m, v, n} (HOST array of 5 x 64-bit words per tensor).
// Tensors of >= 256 Ki elements get their own launch (bandwidth-bound anyway); the small ones -- a transformer has hundreds: LayerNorm
// gains, biases, scales, position-MLP layers -- are packed ADAM_T tensors / ADAM_B blocks per launch (256 per-tensor launches -> ~10).
extern "C" int pk_adamw_multi(const long long* table, int count, float lr, float beta1, float beta2, float eps, float wd, int step, void* stream) {
    if (!table || count <= 0 || step <= 0) return PK_EINVAL;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step), rs = 1.0f / sqrtf(bc2);
    hipStream_t s = STREAM(stream);
    AdamMulti a;
    int nt = 0, nb = 0;
    auto flush = [&]() {
        if (nb > 0) hipLaunchKernelGGL(adamw_multi_kernel, dim3(nb), dim3(256), 0, s, a, lr, beta1, beta2, eps, wd, bc1, rs);
        nt = 0; nb = 0;
    };
    // large tensors: up to ADAM_BIG_T per launch (A/B switch PK_ADAMW_BIG=0: one launch per tensor, the round-3 form)
    static const bool big_multi = !(getenv("PK_ADAMW_BIG") && getenv("PK_ADAMW_BIG")[0] == '0');
    AdamBig big;
    int bt = 0;
    long bblocks = 0;
    auto flush_big = [&]() {
        if (bt > 0) hipLaunchKernelGGL(adamw_big_kernel, dim3((unsigned)bblocks), dim3(256), 0, s, big, bt, lr, beta1, beta2, eps, wd, bc1, rs);
        bt = 0; bblocks = 0;
    };
    for (int i = 0; i < count; ++i) {
        const long long* e = table + 5 * (long)i;
        float* p = reinterpret_cast<float*>(e[0]);
        const float* g = reinterpret_cast<const float*>(e[1]);
        float* m = reinterpret_cast<float*>(e[2]);
        float* v = reinterpret_cast<float*>(e[3]);
        const long n = e[4];
        if (!p || !g || !m || !v || n <= 0) return PK_EINVAL;
        if (n >= 262144) {
            if (!big_multi) {
                hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(n)), dim3(256), 0, s, p, g, m, v, lr, beta1, beta2, eps, wd, bc1, rs, n);
                continue;
            }
            const long nbk = (n + ADAM_BIG_CHUNK - 1) / ADAM_BIG_CHUNK;
            if (bt == ADAM_BIG_T || bblocks + nbk > 0x7fffffffL) flush_big();
            big.p[bt] = p; big.g[bt] = g; big.m[bt] = m; big.v[bt] = v; big.n[bt] = n; big.blk0[bt] = (int)bblocks;
            bblocks += nbk;
            ++bt;
            continue;
        }
        const int chunks = (int)((n + ADAM_CHUNK - 1) / ADAM_CHUNK);
        int c = 0;
        while (c < chunks) {
            if (nt == ADAM_T || nb == ADAM_B) flush();
            a.p[nt] = p; a.g[nt] = g; a.m[nt] = m; a.v[nt] = v; a.n[nt] = (int)n;
            while (c < chunks && nb < ADAM_B) { a.bt[nb] = (unsigned char)nt; a.bc[nb] = (unsigned short)c; ++nb; ++c; }
            ++nt;
        }
    }
    flush();
    flush_big();
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// step = 1, 2, ...: the number of this update (bias corrections 1 - beta^step); wd = 0: plain Adam
extern "C" int pk_adamw(float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps, float wd, int step, long n, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return PK_EINVAL;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(n)), dim3(256), 0, STREAM(stream), p, g, m, v, lr, beta1, beta2, eps, wd, bc1, 1.0f / sqrtf(bc2), n);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
