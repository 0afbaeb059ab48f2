// Cosine-sim multi-head attention of the reference (attention.py:128-182) as two kernels:
//
//  pk_attn_prep : head split, null-kv prepend (interleaved k,v,k,v rows of `null_kv`, attention.py:148),
//                 l2norm of q and of k AFTER the null-k concat (attention.py:153), * q_scale / k_scale,
//                 * scale (8, attention.py:157, folded into q), V stored TRANSPOSED per head.
//                 Layouts written (T = bf16 | f32):
//                   Qp [S][h][nq_pad][64]   Kp [S][h][nk_pad][64]   Vt [S][h][64][nk_pad]
//                 nk = nnull + n_kv, pads are zero-filled.
//  pk_attn_fwd  : softmax(Qp Kp^T + bias (+ALiBi, causal, key mask)) V, flash-style, LDS-free:
//                 one wave owns 16*QF query rows; per 32-key tile it computes S^T = K Q^T and
//                 O^T = V^T P^T with MFMA, so every softmax statistic of query row (lane & 15) is
//                 lane-local up to a 4-lane-group shuffle, and P never leaves registers
//                 (the S^T accumulator layout IS the P^T operand layout under the key permutation
//                 key = (j >> 2)*16 + g*4 + (j & 3) that V^T's fragment loads use too).
// dim_head is fixed at 64 (the reference default; every BASELINE config).
// Roofline: MFMA for n = 576 (4*nq*nk*64 flops per head), L1/L2 operand-fetch bound at small QF.
#include <cstdlib>
#include "common.hpp"

namespace pk {

constexpr int DH = 64;

struct PrepArgs {
    const float* q; int ldq;        // [S*nq][ldq], head hh at columns hh*64
    const float* kv; int ldkv;      // [S*n_kv][ldkv], k at columns [0, h*64), v at [h*64, 2*h*64)
    const float* null_kv;           // [h][2*nnull][64] or null
    const float* q_scale;           // [64]
    const float* k_scale;           // [64]
    void* Qp; void* Kp; void* Vt;
    int S, h, nq, n_kv, nnull, nq_pad, nk_pad;
    float scale;
};

// 16 lanes per (sequence, head, row): each lane owns 4 consecutive head dims (one 16-byte load), the row's
// sum of squares is a 4-step xor-shuffle inside the 16-lane group.
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}

template <typename T>
__device__ __forceinline__ void prep_q_block(const PrepArgs& p, long bid) {
    const int l16 = threadIdx.x & 15;
    const long r = bid * 16 + (threadIdx.x >> 4);                        // row index over (s, hh, i) incl. pad rows
    const long total = (long)p.S * p.h * p.nq_pad;
    if (r >= total) return;
    const int i = (int)(r % p.nq_pad);
    const long sh = r / p.nq_pad;
    const int hh = (int)(sh % p.h), s = (int)(sh / p.h);
    f32x4 v = f32x4{0, 0, 0, 0};
    if (i < p.nq) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p.q + ((size_t)s * p.nq + i) * p.ldq + hh * DH + l16 * 4);
        if (p.q_scale) {
            const float ss = group16_sum((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]));
            const float inv = p.scale / fmaxf(sqrtf(ss), 1e-12f);               // F.normalize eps = 1e-12
            const f32x4 qs = *reinterpret_cast<const f32x4*>(p.q_scale + l16 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = x[e] * inv * qs[e];
        } else {                                                                // plain dot-product attention (T5): q * scale, no l2norm
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = x[e] * p.scale;
        }
    }
    store4(reinterpret_cast<T*>(p.Qp) + (size_t)r * DH + l16 * 4, v);
}

template <typename T>
__device__ __forceinline__ void prep_kv_block(const PrepArgs& p, float (*vt)[65], int bid) {
    // one block per (s, hh, 64-key tile); thread -> (key = tid >> 4 (+16 per pass), 4 head dims)
    const int l16 = threadIdx.x & 15;
    const int tiles = (p.nk_pad + 63) / 64;
    const int kt = bid % tiles;
    const int sh = bid / tiles;
    const int hh = sh % p.h, s = sh / p.h;
    const int nk = p.nnull + p.n_kv;
    T* Kp = reinterpret_cast<T*>(p.Kp) + (size_t)sh * p.nk_pad * DH;
    T* Vt = reinterpret_cast<T*>(p.Vt) + (size_t)sh * DH * p.nk_pad;
    const int kend = (p.nk_pad - kt * 64) < 64 ? (p.nk_pad - kt * 64) : 64;   // keys of this tile that exist in the padded image
    const f32x4 ks = p.k_scale ? *reinterpret_cast<const f32x4*>(p.k_scale + l16 * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
    for (int jj = threadIdx.x >> 4; jj < kend; jj += 16) {
        const int key = kt * 64 + jj;
        f32x4 kx = f32x4{0, 0, 0, 0}, vx = kx;
        if (key < p.nnull) {
            kx = *reinterpret_cast<const f32x4*>(p.null_kv + ((size_t)hh * 2 * p.nnull + 2 * key) * DH + l16 * 4);
            vx = *reinterpret_cast<const f32x4*>(p.null_kv + ((size_t)hh * 2 * p.nnull + 2 * key + 1) * DH + l16 * 4);
        } else if (key < nk) {
            const float* row = p.kv + ((size_t)s * p.n_kv + (key - p.nnull)) * p.ldkv;
            kx = *reinterpret_cast<const f32x4*>(row + hh * DH + l16 * 4);
            vx = *reinterpret_cast<const f32x4*>(row + p.h * DH + hh * DH + l16 * 4);
        }
        const float ss = group16_sum((kx[0] * kx[0] + kx[1] * kx[1]) + (kx[2] * kx[2] + kx[3] * kx[3]));
        const float inv = key < nk ? (p.k_scale ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 1.0f) : 0.f;      // k_scale == NULL: plain keys (T5)
        f32x4 kn;
#pragma unroll
        for (int e = 0; e < 4; ++e) { kn[e] = kx[e] * inv * ks[e]; vt[jj][l16 * 4 + e] = vx[e]; }
        store4(Kp + (size_t)key * DH + l16 * 4, kn);
    }
    __syncthreads();
    // transposed store: thread -> (d = tid >> 2, 16 consecutive keys)
    const int d = threadIdx.x >> 2, j0 = (threadIdx.x & 3) * 16;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int jl = j0 + q4 * 4;
        if (jl < kend) {   // nk_pad % 4 == 0
            const f32x4 o = f32x4{vt[jl + 0][d], vt[jl + 1][d], vt[jl + 2][d], vt[jl + 3][d]};
            store4(Vt + (size_t)d * p.nk_pad + kt * 64 + jl, o);
        }
    }
}

// both operand-image kernels in ONE launch (round 6): blocks [0, nqb) lay down the Q^ image, the rest the K^ / V^T images
template <typename T>
__global__ __launch_bounds__(256) void prep_qkv_kernel(const PrepArgs p, int nqb) {
    __shared__ float vt[64][65];
    if ((int)blockIdx.x < nqb) prep_q_block<T>(p, blockIdx.x);
    else prep_kv_block<T>(p, vt, (int)blockIdx.x - nqb);
}

struct AttnArgs {
    const void* Qp; const void* Kp; const void* Vt;
    const float* bias; long bias_hstride; int bias_ld;   // bias[hh][i][j] over REAL keys j, or null
    const unsigned char* kmask;                            // [S][n_kv] (1 = keep) over real keys, or null
    const float* slopes;                                   // ALiBi slopes [h] (causal layers), or null
    void* O; int ldo; int out_f32;                         // O[(s*nq + i)][hh*64 + d]
    int S, h, nq, n_kv, nnull, nq_pad, nk_pad, causal;
    int bias_vec;                                          // bias rows are 16-byte loadable (nnull == 0, aligned strides)
    // relative-position bias as a TABLE (LDS-staged kernel only): bias[hh][i][j] = bias_tab[hh][pos_code[i] - pos_code[j] + code_off].
    // The continuous position bias of the reference (attention.py:229-275) depends on (i, j) only through the relative grid
    // position: (2T-1)(2H-1)(2W-1) = 3825 distinct values per head at (9, 8, 8) -- 15 KB in LDS instead of a 1.33 MB f32 stream per
    // (sequence, head) through L2 (170 MB per launch at 16 x 8: the n = 576 kernel ran 37 us with the stream, 9 us without a bias)
    const float* bias_tab; int tab_len; const int* pos_code; int code_off;
    int tab_run4;                                          // pos_code[4k + r] == pos_code[4k] + r (last grid dimension a multiple of 4)
    // fixed-offset softmax (LDS-staged kernel, no key mask / causal): the caller knows an upper bound of sim + bias (|sim| <= scale *
    // max|q_scale . k_scale| because q^ and k^ are unit vectors; the bias table has a known maximum), so p = 2^(s*log2e - off2) with
    // the INTEGER off2 = ceil(bound * log2e) can never overflow and no running maximum, no cross-lane max, no accumulator rescale is
    // needed: ~40 % of the VALU work of a key tile.  An integer shift of the exponent leaves every mantissa -- hence the bf16
    // rounding of p -- independent of the tiling.  off2 < 0: running-max softmax.
    float off2;
    // training forward (LDS-free kernel only): lse[(s h + hh) nq + i] = log sum_j exp(score[i][j]) for the backward kernels (round 6: they recomputed
    // it with one extra pass over the keys); null: not written
    float* lse;
};

constexpr float ATTN_LOG2E = 1.4426950408889634f;
#ifdef PK_TIMELINE
// instrumented build only (tools/build_alt.sh with PK_ALT_SRC=attn; tools/attn_timeline.py): s_memtime stamps of wave 0 of a few workgroups
__device__ unsigned long long pk_attn_tl[8 * 6 * 16];
#define PK_ATL(slot) do { if (tl_on && t < 14) pk_attn_tl[(tl_wg * 16 + t) * 6 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define PK_ATL_K(slot) do { if (tl_on) pk_attn_tl[(tl_wg * 16 + 15) * 6 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PK_ATL(slot) do {} while (0)
#define PK_ATL_K(slot) do {} while (0)
#endif
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));      // 8-byte LDS read at dword alignment (ds_read2_b32)

__device__ __forceinline__ void load_vt(Frag<bf16>& f, const bf16* row, int kb, int g) {
    const u32x2 a = *reinterpret_cast<const u32x2*>(row + kb + g * 4);
    const u32x2 b = *reinterpret_cast<const u32x2*>(row + kb + 16 + g * 4);
    f.v = u32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ void load_vt(Frag<float>& f, const float* row, int kb, int g) {
    f.lo = *reinterpret_cast<const f32x4*>(row + kb + g * 4);
    f.hi = *reinterpret_cast<const f32x4*>(row + kb + 16 + g * 4);
}

// pre-split image (common.hpp bf16x3p): keys kb + g*4 .. +3 and kb + 16 + g*4 .. +3 of one V^T row, from both planes of the block
__device__ __forceinline__ void load_vt(Frag<bf16x3p>& f, const bf16x3p* row, int kb, int g) {
    int o;
    const char* blk = split_block(row + kb, o);               // kb % 32 == 0 and rows start on a block: o == 0
    const u32x2 a = *reinterpret_cast<const u32x2*>(blk + o + g * 8), b = *reinterpret_cast<const u32x2*>(blk + o + 32 + g * 8);
    const u32x2 c = *reinterpret_cast<const u32x2*>(blk + 64 + o + g * 8), d = *reinterpret_cast<const u32x2*>(blk + 64 + o + 32 + g * 8);
    f.hi = u32x4{a[0], a[1], b[0], b[1]};
    f.lo = u32x4{c[0], c[1], d[0], d[1]};
}

// V^T columns of keys >= nk (tile padding) are multiplied by p = 0 exactly, but 0 * NaN = NaN: the images written by
// pk_qkv_project leave those columns untouched, so the tail tile's fragment is masked here instead of zero-filling
// 17 MB of V^T with a separate launch per layer.  Element e of a fragment is key kb + (e >> 2) * 16 + g * 4 + (e & 3).
__device__ __forceinline__ void mask_vt_tail(Frag<bf16>& f, int kb, int g, int nk) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int key = kb + (w >> 1) * 16 + g * 4 + (w & 1) * 2;
        const uint32_t m = (key < nk ? 0x0000FFFFu : 0u) | (key + 1 < nk ? 0xFFFF0000u : 0u);
        f.v[w] &= m;
    }
}
__device__ __forceinline__ void mask_vt_tail(Frag<bf16x3p>& f, int kb, int g, int nk) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int key = kb + (w >> 1) * 16 + g * 4 + (w & 1) * 2;
        const uint32_t m = (key < nk ? 0x0000FFFFu : 0u) | (key + 1 < nk ? 0xFFFF0000u : 0u);
        f.hi[w] &= m; f.lo[w] &= m;
    }
}
__device__ __forceinline__ void mask_vt_tail(Frag<float>& f, int kb, int g, int nk) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (kb + g * 4 + e >= nk) f.lo[e] = 0.f;
        if (kb + 16 + g * 4 + e >= nk) f.hi[e] = 0.f;
    }
}

template <typename T, int QF>
// QF = 2 on f32 / split-bf16 operands (the training step's n = 576 forward, the f32 / bf16x3 parity modes): the compiler's free choice was
// ~250 VGPRs + 32 AGPRs = one wave per SIMD, 256 resident workgroups for the 320 of a B = 8 call; two waves per SIMD put them in one round
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((QF == 2 && sizeof(T) != 2) ? 2 : 1))) void attn_fwd_kernel(const AttnArgs p) {
    const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
    const int qtiles = p.nq_pad / (16 * QF);
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long)p.S * p.h * qtiles) return;
    const int qt = (int)(wid % qtiles);
    const int sh = (int)(wid / qtiles);
    const int hh = sh % p.h, s = sh / p.h;
    const int q0 = qt * 16 * QF;
    const int nk = p.nnull + p.n_kv;

    const T* Qp = reinterpret_cast<const T*>(p.Qp) + ((size_t)sh * p.nq_pad + q0) * DH;
    const T* Kp = reinterpret_cast<const T*>(p.Kp) + (size_t)sh * p.nk_pad * DH;
    const T* Vt = reinterpret_cast<const T*>(p.Vt) + (size_t)sh * DH * p.nk_pad;

    Frag<T> fq[QF][2];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf)
#pragma unroll
        for (int c = 0; c < 2; ++c) frag_load(fq[qf][c], Qp + (size_t)(qf * 16 + lr) * DH + c * 32 + g * 8);

    float m[QF], l[QF];
    f32x4 o[QF][4];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        m[qf] = -INFINITY; l[qf] = 0.f;
#pragma unroll
        for (int df = 0; df < 4; ++df) o[qf][df] = f32x4{0, 0, 0, 0};
    }
    const float slope = (p.causal && p.slopes) ? p.slopes[hh] : 0.f;
    const float* bias = p.bias ? p.bias + (size_t)hh * p.bias_hstride : nullptr;
    const unsigned char* km = p.kmask ? p.kmask + (size_t)s * p.n_kv : nullptr;
    const int coff = p.n_kv - p.nq;                         // causal diagonal offset (attention.py:172)

    // software pipeline: the K fragments (and, on the vector-bias path, the bias vectors) of tile t+1 are requested
    // before tile t's MFMAs and softmax, V^T of tile t at the top of the iteration -- the loop is operand-fetch bound
    // (every wave streams K / V^T / bias through L1), so the loads must not sit behind the dependent MFMA chain
    const bool vb_all = bias && p.bias_vec && !km && !p.causal;           // wave-uniform, loop-invariant
    int qrow[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) { const int qi = q0 + qf * 16 + lr; qrow[qf] = qi < p.nq ? qi : p.nq - 1; }
    auto load_k = [&](Frag<T> (&fk)[2][2], int kb) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int c = 0; c < 2; ++c) frag_load(fk[f][c], Kp + (size_t)(kb + f * 16 + lr) * DH + c * 32 + g * 8);
    };
    auto load_bias = [&](f32x4 (&bz)[QF][2], int kb) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                bz[qf][f] = *reinterpret_cast<const f32x4*>(bias + (size_t)qrow[qf] * p.bias_ld + kb + f * 16 + g * 4);
    };
    Frag<T> fk[2][2];
    f32x4 bz[QF][2];
    load_k(fk, 0);
    if (vb_all && 32 <= nk) load_bias(bz, 0);

    for (int kb = 0; kb < p.nk_pad; kb += 32) {
        Frag<T> fv[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) load_vt(fv[df], Vt + (size_t)(df * 16 + lr) * p.nk_pad, kb, g);
        if (kb + 32 > nk) {                                   // wave-uniform: only the tail tile
#pragma unroll
            for (int df = 0; df < 4; ++df) mask_vt_tail(fv[df], kb, g, nk);
        }
        Frag<T> fkn[2][2];
        f32x4 bzn[QF][2];
        const bool has_next = kb + 32 < p.nk_pad;
        if (has_next) {
            load_k(fkn, kb + 32);
            if (vb_all && kb + 64 <= nk) load_bias(bzn, kb + 32);
        }
        f32x4 st[QF][2];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) { st[qf][0] = f32x4{0, 0, 0, 0}; st[qf][1] = st[qf][0]; }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) st[qf][f] = mma(fk[f][c], fq[qf][c], st[qf][f]);
        // whole-tile fast paths (wave-uniform): a full tile of real keys with no key mask / causal structure needs no
        // per-element work; the additive bias of such a tile is one 16-byte load per 4 keys (nnull == 0, aligned rows)
        const bool simple = (kb + 32 <= nk) && !km && !p.causal;
        const bool vbias = vb_all && simple;
        const bool plain = simple && (!bias || vbias);
        float pr[QF][8];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const int qi = q0 + qf * 16 + lr;
            float mx = -INFINITY;
            if (vbias) {
#pragma unroll
                for (int f = 0; f < 2; ++f) st[qf][f] += bz[qf][f];
            }
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sv = st[qf][f][r];
                    if (!plain) {
                        const int key = kb + f * 16 + g * 4 + r;
                        const int j = key - p.nnull;
                        if (key >= nk) sv = -INFINITY;               // tile padding: contributes exactly 0
                        else {
                            if (bias && j >= 0 && qi < p.nq) sv += bias[(size_t)qi * p.bias_ld + j];
                            bool masked = km && j >= 0 && !km[j];
                            if (p.causal) {                          // ALiBi runs over the null keys too (j < 0: never masked), attention.py:198-227
                                const int dj = j - (qi + coff);
                                sv -= fabsf((float)dj) * slope;
                                masked = masked || dj > 0;
                            }
                            if (masked) sv = NEG_MAX;
                        }
                    }
                    pr[qf][f * 4 + r] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[qf], mx);               // finite: every 32-key tile holds >= 1 real key
            const float alpha = __expf(m[qf] - mn);
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { pr[qf][e] = __expf(pr[qf][e] - mn); ls += pr[qf][e]; }
            l[qf] = l[qf] * alpha + ls;
            m[qf] = mn;
#pragma unroll
            for (int df = 0; df < 4; ++df) o[qf][df] *= alpha;
        }
        Frag<T> fp[QF];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) frag_from_f32(fp[qf], pr[qf]);
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) o[qf][df] = mma(fv[df], fp[qf], o[qf][df]);
        if (has_next) {
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int c = 0; c < 2; ++c) fk[f][c] = fkn[f][c];
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) { bz[qf][0] = bzn[qf][0]; bz[qf][1] = bzn[qf][1]; }
        }
    }
    float* Of = reinterpret_cast<float*>(p.O);
    T* Ot = reinterpret_cast<T*>(p.O);
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        float lt = l[qf];
        lt += __shfl_xor(lt, 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const float inv = 1.0f / lt;
        const int qi = q0 + qf * 16 + lr;
        if (qi < p.nq) {
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                const size_t off = ((size_t)s * p.nq + qi) * p.ldo + hh * DH + df * 16 + g * 4;
                const f32x4 v = o[qf][df] * inv;
                if (p.out_f32) store4(Of + off, v); else store4(Ot + off, v);
            }
            if (p.lse && g == 0) p.lse[(size_t)sh * p.nq + qi] = m[qf] + __logf(lt);
        }
    }
}


// LDS fragment reads of the staged tiles (see attn_fwd_lds_kernel).
// The keys of a 64-key tile are fed to the QK MFMAs in a PERMUTED order: row i = 4*g' + r' of 16-key block f is key
//   kperm(f, i) = (f >> 1)*32 + g'*8 + (f & 1)*4 + r'
// so that after S^T = K Q^T lane group g holds, for each 32-key chunk kc, the EIGHT CONSECUTIVE keys kc*32 + g*8 + 0..7
// (blocks f = 2kc and 2kc+1, r = 0..3).  The P^T operand of the PV MFMA then pairs with 16 contiguous bytes of a V^T
// row: ONE ds_read_b128 per fragment instead of two 8-byte pieces 32 B apart (which the compiler fused into
// ds_read2_b64 -- 16-lane groups over 32 banks -- and which conflicted 2-way: SQ_LDS_BANK_CONFLICT was 38 % of
// SQ_LDS_IDX_ACTIVE).  The contraction order over keys is irrelevant as long as P^T and V^T agree.
// K tile [64 keys][128 B]: 16-B slot ^ ksw(row), ksw chosen so the permuted rows of every ds_read_b128 lane group hit
// distinct banks; V^T tile [64 dims][128 B = 64 keys]: slot ^ (row & 7) as in the GEMM.
__device__ __forceinline__ int attn_kperm(int f, int i) { return (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3); }
__device__ __forceinline__ int attn_ksw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }
__device__ __forceinline__ void lds_frag_k(Frag<bf16>& f, const char* tile, int row, int chunk, int g) {
    const int slot = (chunk * 4 + g) ^ attn_ksw(row);
    f.v = *reinterpret_cast<const u32x4*>(tile + row * 128 + (slot << 4));
}
__device__ __forceinline__ void lds_frag_vt(Frag<bf16>& f, const char* tile, int row, int kc, int g) {
    const int slot = (kc * 4 + g) ^ (row & 7);              // keys kc*32 + g*8 + 0..7 of head dim `row`
    f.v = *reinterpret_cast<const u32x4*>(tile + row * 128 + (slot << 4));
}

// split-bf16 images (common.hpp bf16x3p): a 64-element row is two 128-byte blocks [hi x 32 | lo x 32]; block c of every row of a tile forms a
// sub-tile [64 rows][128 B] with the same slot swizzle as the bf16 tile, hi in slots 0..3, lo in slots 4..7 (tile = the sub-tile of chunk c)
__device__ __forceinline__ void lds_frag_k(Frag<bf16x3p>& f, const char* tile, int row, int chunk, int g) {
    const char* base = tile + chunk * 8192 + row * 128;
    f.hi = *reinterpret_cast<const u32x4*>(base + ((g ^ attn_ksw(row)) << 4));
    f.lo = *reinterpret_cast<const u32x4*>(base + (((4 + g) ^ attn_ksw(row)) << 4));
}
__device__ __forceinline__ void lds_frag_vt(Frag<bf16x3p>& f, const char* tile, int row, int kc, int g) {
    const char* base = tile + kc * 8192 + row * 128;
    f.hi = *reinterpret_cast<const u32x4*>(base + ((g ^ (row & 7)) << 4));
    f.lo = *reinterpret_cast<const u32x4*>(base + (((4 + g) ^ (row & 7)) << 4));
}
__device__ __forceinline__ void frag_ones(Frag<bf16>& f) { f.v = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; }
__device__ __forceinline__ void frag_ones(Frag<bf16x3p>& f) { f.hi = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; f.lo = u32x4{0, 0, 0, 0}; }

// ---- LDS-staged variant (bf16; T = bf16x3p: the split-bf16 images, tiles twice as large): one workgroup = 4 waves = 64*QF query rows of ONE (sequence, head).  K and V^T tiles of
// 64 keys are fetched once per workgroup in full 128-byte lines by LDS-DMA (buffer_load ... lds) into a 2-stage ring
// and read back as MFMA fragments with ds_read_b128 / ds_read_b64 (XOR-swizzled on the DMA source side, conflict-free),
// instead of every wave pulling fragment-shaped pieces (16 rows x 64 B per instruction) through the texture path.
typedef __attribute__((address_space(3))) void* attn_lds_ptr;

// The K / V^T ring has 2 stages: 3- and 4-stage rings behind a counted vmcnt were measured slower (35.1 / 43.3 vs 32.7 us at
// n = 576, profiles/attn_variants_r02.txt -- the loop is bound by its VALU stream, not by DMA latency) and were removed.
// FIX: fixed-offset softmax (AttnArgs::off2), host-selected when there is no key mask and no causal mask.
template <int QF, bool PF, bool TAB = false, bool FIX = false, typename T = bf16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? 1 : (QF == 1 ? (PF ? 4 : (TAB ? 3 : 5)) : ((FIX && !TAB) ? 4 : 3))))) void attn_fwd_lds_kernel(const AttnArgs p, uint32_t kv_bytes) {
    constexpr int STAGES = 2;
    constexpr bool X3 = sizeof(T) == 4;                                   // split-bf16 image: 4 bytes per element, two 128-byte blocks per 64-element row
    constexpr int NB = X3 ? 2 : 1;                                        // 128-byte blocks per tile row
    constexpr int KT = 8192 * NB;                                         // bytes of a K (or V^T) tile of 64 keys
#ifdef PK_TIMELINE
    const int tl_wg = blockIdx.x == 0 ? 0 : blockIdx.x == 8 ? 1 : blockIdx.x == 1 ? 2 : blockIdx.x == gridDim.x / 2 ? 3 :
                      blockIdx.x == gridDim.x - 8 ? 4 : blockIdx.x == 256 ? 5 : blockIdx.x == 300 ? 6 : blockIdx.x == 511 ? 7 : -1;
    const bool tl_on = tl_wg >= 0 && threadIdx.x == 0;
    PK_ATL_K(0);
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];          // STAGES x (K 8 KB | V^T 8 KB) [| bias table | position codes]
    constexpr int STAGE = 2 * KT;
    float* tab = reinterpret_cast<float*>(smem + STAGES * STAGE);         // TAB: this head's bias table, then the position codes of all keys
    int* codes = reinterpret_cast<int*>(smem + STAGES * STAGE + ((p.tab_len * 4 + 15) & ~15));
    const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qblocks = (p.nq_pad + 64 * QF - 1) / (64 * QF);
    const int qb = blockIdx.x % qblocks;
    const int sh = blockIdx.x / qblocks;
    const int hh = sh % p.h, s = sh / p.h;
    const int q0 = (qb * 4 + wave) * 16 * QF;
    const bool active = q0 < p.nq_pad;                                    // wave-uniform; idle waves still feed the ring
    const int nk = p.nnull + p.n_kv;

    __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Kp), 0, kv_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Vt), 0, kv_bytes, 0x00020000);
    // DMA pieces: 8 rows x 128 B each.  K tile: rows = keys (row stride 128 B); V^T tile: rows = head dims (row stride
    // nk_pad * 2 B).  Each wave issues 2 K pieces + 2 V^T pieces per tile.
    const int prow = lane >> 3, pslot = lane & 7;
    uint32_t offK[2], offV[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + prow;                        // 0..63
        offK[i] = ((uint32_t)sh * p.nk_pad + row) * (128u * NB) + (uint32_t)((pslot ^ attn_ksw(row)) * 16);
        offV[i] = ((uint32_t)sh * 64u + row) * (uint32_t)p.nk_pad * (2u * NB) + (uint32_t)((pslot ^ (row & 7)) * 16);
    }
    auto issue = [&](int kb, int stage) {
        char* base = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < NB; ++c) {                                 // block c of the rows -> sub-tile c
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (attn_lds_ptr)(base + c * 8192 + (wave * 2 + i) * 1024), 16, offK[i] + c * 128, kb * 128 * NB, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (attn_lds_ptr)(base + KT + c * 8192 + (wave * 2 + i) * 1024), 16, offV[i] + c * 128, kb * 2 * NB, 0, 0);
            }
    };

    const T* Qp = reinterpret_cast<const T*>(p.Qp) + ((size_t)sh * p.nq_pad + (active ? q0 : 0)) * DH;
    Frag<T> fq[QF][2];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf)
#pragma unroll
        for (int c = 0; c < 2; ++c) frag_load(fq[qf][c], Qp + (size_t)(qf * 16 + lr) * DH + c * 32 + g * 8);

    float m[QF], l[QF];
    f32x4 o[QF][4], lsum[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        m[qf] = -INFINITY; l[qf] = 0.f; lsum[qf] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int df = 0; df < 4; ++df) o[qf][df] = f32x4{0, 0, 0, 0};
    }
    const float slope = (p.causal && p.slopes) ? p.slopes[hh] : 0.f;
    const float* bias = (p.bias && !FIX) ? p.bias + (size_t)hh * p.bias_hstride : nullptr;
    const unsigned char* km = (p.kmask && !FIX) ? p.kmask + (size_t)s * p.n_kv : nullptr;
    const int coff = p.n_kv - p.nq;
    const bool vb_all = bias && p.bias_vec && !km && !p.causal;
    int qrow[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) { const int qi = q0 + qf * 16 + lr; qrow[qf] = qi < p.nq ? qi : p.nq - 1; }
    int cq[QF];                                                            // TAB: position code of the lane's query row(s) + offset
    if (TAB) {
        // staging loads are issued in batches of 8 independent requests per thread: written as one load -> store per iteration the
        // 15 trips of the 3825-entry table were 15 dependent L2 / HBM round trips -- 11.4 k of the kernel's 46 k cycles (s_memtime
        // timeline, tools/attn_timeline.py) before the first key tile
        const float* src = p.bias_tab + (size_t)hh * p.tab_len;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) cq[qf] = p.pos_code[qrow[qf]] + p.code_off;
        for (int i0 = threadIdx.x; i0 < p.tab_len; i0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; v[u] = i < p.tab_len ? src[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; if (i < p.tab_len) tab[i] = FIX ? fmaf(v[u], ATTN_LOG2E, -p.off2) : v[u]; }
        }
        for (int i0 = threadIdx.x; i0 < p.n_kv; i0 += 4 * 256) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; v[u] = i < p.n_kv ? p.pos_code[i] : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; if (i < p.n_kv) codes[i] = v[u]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // visible to every wave after the first barrier of the loop
    }

    const int ntiles = (p.nk_pad + 63) / 64;
    // additive-bias vectors (16 B per lane per 4 keys, straight from global) are requested ONE TILE AHEAD: used right
    // after the QK MFMAs of their own tile they left an L2 round trip exposed in every one of the nk/64 iterations
    auto load_bz = [&](f32x4 (&b)[QF][4], int kb) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                b[qf][f] = *reinterpret_cast<const f32x4*>(bias + (size_t)qrow[qf] * p.bias_ld + kb + attn_kperm(f, g * 4));
    };
    f32x4 bz[QF][4];
    if (PF && active && vb_all && 64 <= nk) load_bz(bz, 0);
#pragma unroll
    for (int s0 = 0; s0 < STAGES - 1; ++s0)
        if (s0 < ntiles) issue(s0 * 64, s0);
    for (int t = 0; t < ntiles; ++t) {
        const int kb = t * 64;
        PK_ATL(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PK_ATL(1);
        __builtin_amdgcn_s_barrier();                     // tile t landed for all waves; everyone finished tile t-1
        PK_ATL(2);
        if (t + STAGES - 1 < ntiles) issue(kb + (STAGES - 1) * 64, (t + STAGES - 1) % STAGES);
        PK_ATL(3);
        if (kb + 64 > nk) {
            // tail tile (wave-uniform, at most once per workgroup): the V^T columns of keys >= nk carry p = 0 but may hold anything
            // (pk_qkv_project never writes them; 0 * NaN = NaN) -> zero them in the LDS image, all 256 threads, one extra barrier.
            // (Masking the fragments in registers instead cost 13 VGPRs and spilled the main loop.)
            char* vz = smem + (t % STAGES) * STAGE + KT;
            const int d = threadIdx.x >> 2, k0 = (threadIdx.x & 3) * 16;
            for (int kk = k0; kk < k0 + 16; ++kk)
                if (kb + kk >= nk) {
                    if (X3) {                                              // key kk: sub-tile kk >> 5, position kk & 31 in both planes of the block
                        char* blk = vz + (kk >> 5) * 8192 + d * 128;
                        const int pos = kk & 31;
                        *reinterpret_cast<u16*>(blk + ((((pos >> 3)) ^ (d & 7)) << 4) + (pos & 7) * 2) = 0;
                        *reinterpret_cast<u16*>(blk + (((4 + (pos >> 3)) ^ (d & 7)) << 4) + (pos & 7) * 2) = 0;
                    } else {
                        *reinterpret_cast<u16*>(vz + d * 128 + ((((kk >> 3) ^ (d & 7))) << 4) + (kk & 7) * 2) = 0;
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (!active) continue;
        const char* kt = smem + (t % STAGES) * STAGE;
        const char* vt = kt + KT;
        const bool simple = (kb + 64 <= nk) && !km && !p.causal;
        const bool vbias = vb_all && simple;
        const bool plain = simple && (!bias || vbias);
        const bool vbias_next = PF && vb_all && t + 1 < ntiles && kb + 128 <= nk;
        f32x4 bzn[QF][4];
        if (vbias_next) load_bz(bzn, kb + 64);
        if (!PF && vbias) load_bz(bz, kb);
        f32x4 st[QF][4];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf)
#pragma unroll
            for (int f = 0; f < 4; ++f) st[qf][f] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                Frag<T> fk;
                lds_frag_k(fk, kt, attn_kperm(f, lr), c, g);
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) st[qf][f] = mma(fk, fq[qf][c], st[qf][f]);
            }
        float pr[QF][16];
        PK_ATL(4);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            const int qi = q0 + qf * 16 + lr;
            float mx = -INFINITY;
            if (vbias) {
#pragma unroll
                for (int f = 0; f < 4; ++f) st[qf][f] += bz[qf][f];
            }
            if (FIX) {
                // p = 2^(s * log2e + bias * log2e - off2): the table already holds bias * log2e - off2 (host: no other bias form here)
                const float noff = -p.off2;
                if (TAB && simple && p.tab_run4) {
                    // the lane's 4 keys of block f are consecutive AND so are their position codes: their table entries are 4
                    // consecutive floats (descending in r) -> one code read + two dword-aligned 8-byte LDS reads instead of 4 gathers
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const float* tp = tab + (cq[qf] - codes[kb + attn_kperm(f, g * 4)] - 3);
                        const f32x2u lo = *reinterpret_cast<const f32x2u*>(tp), hi = *reinterpret_cast<const f32x2u*>(tp + 2);
                        pr[qf][f * 4 + 0] = __builtin_amdgcn_exp2f(fmaf(st[qf][f][0], ATTN_LOG2E, hi[1]));
                        pr[qf][f * 4 + 1] = __builtin_amdgcn_exp2f(fmaf(st[qf][f][1], ATTN_LOG2E, hi[0]));
                        pr[qf][f * 4 + 2] = __builtin_amdgcn_exp2f(fmaf(st[qf][f][2], ATTN_LOG2E, lo[1]));
                        pr[qf][f * 4 + 3] = __builtin_amdgcn_exp2f(fmaf(st[qf][f][3], ATTN_LOG2E, lo[0]));
                    }
                } else if (TAB && simple) {
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const u32x4 ck = *reinterpret_cast<const u32x4*>(codes + kb + attn_kperm(f, g * 4));
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            pr[qf][f * 4 + r] = __builtin_amdgcn_exp2f(fmaf(st[qf][f][r], ATTN_LOG2E, tab[cq[qf] - (int)ck[r]]));
                    }
                } else {
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float add = noff;
                            bool pad = false;
                            if (!simple) {
                                const int key = kb + attn_kperm(f, g * 4 + r);
                                const int j = key - p.nnull;
                                pad = key >= nk;
                                if (!pad && TAB && j >= 0 && qi < p.nq) add = tab[cq[qf] - codes[j]];
                            }
                            // tile padding is masked by SELECTING 0, never through arithmetic: the K^ rows of keys >= nk are not
                            // written by pk_qkv_project (torch.empty), so their scores may be NaN / Inf and fma(NaN, c, -inf) = NaN
                            const float e2 = __builtin_amdgcn_exp2f(fmaf(st[qf][f][r], ATTN_LOG2E, add));
                            pr[qf][f * 4 + r] = pad ? 0.f : e2;
                        }
                }
                continue;
            }
            if (TAB && simple) {
                // whole tile of real keys: the lane's 4 keys of block f are consecutive -> their position codes are ONE 16-byte LDS read
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const u32x4 ck = *reinterpret_cast<const u32x4*>(codes + kb + attn_kperm(f, g * 4));
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[qf][f][r] += tab[cq[qf] - (int)ck[r]];
                }
            }
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sv = st[qf][f][r];
                    if (!(plain || (TAB && simple))) {
                        const int key = kb + attn_kperm(f, g * 4 + r);
                        const int j = key - p.nnull;
                        if (key >= nk) sv = -INFINITY;
                        else {
                            if (TAB) { if (j >= 0 && qi < p.nq) sv += tab[cq[qf] - codes[j]]; }
                            else if (bias && j >= 0 && qi < p.nq) sv += bias[(size_t)qi * p.bias_ld + j];
                            bool masked = km && j >= 0 && !km[j];
                            if (p.causal) {                          // ALiBi runs over the null keys too (j < 0: never masked), attention.py:198-227
                                const int dj = j - (qi + coff);
                                sv -= fabsf((float)dj) * slope;
                                masked = masked || dj > 0;
                            }
                            if (masked) sv = NEG_MAX;
                        }
                    }
                    pr[qf][f * 4 + r] = sv;
                    mx = fmaxf(mx, sv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m[qf], mx);               // finite: every 64-key tile holds >= 1 real key
            const float alpha = __expf(m[qf] - mn);
            float ls = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { pr[qf][e] = __expf(pr[qf][e] - mn); ls += pr[qf][e]; }
            l[qf] = l[qf] * alpha + ls;
            m[qf] = mn;
#pragma unroll
            for (int df = 0; df < 4; ++df) o[qf][df] *= alpha;
        }
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            Frag<T> fp[QF];
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                const float (&src)[16] = pr[qf];
                const float p8[8] = {src[kc * 8 + 0], src[kc * 8 + 1], src[kc * 8 + 2], src[kc * 8 + 3], src[kc * 8 + 4], src[kc * 8 + 5], src[kc * 8 + 6], src[kc * 8 + 7]};
                frag_from_f32(fp[qf], p8);
            }
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                Frag<T> fv;
                lds_frag_vt(fv, vt, df * 16 + lr, kc, g);
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) o[qf][df] = mma(fv, fp[qf], o[qf][df]);
            }
            if (FIX) {
                // row sums on the matrix core: a V^T block of ones gives l = sum_j bf16(p_j) in every lane of the query row -- the SAME
                // rounded weights the numerator uses (a dominant key contributes exactly p / p), no VALU adds, no cross-lane fold
                Frag<T> ones;
                frag_ones(ones);
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) lsum[qf] = mma(ones, fp[qf], lsum[qf]);
            }
        }
        if (vbias_next) {
#pragma unroll
            for (int qf = 0; qf < QF; ++qf)
#pragma unroll
                for (int f = 0; f < 4; ++f) bz[qf][f] = bzn[qf][f];
        }
        PK_ATL(5);
    }
    PK_ATL_K(1);
    if (!active) return;
    float* Of = reinterpret_cast<float*>(p.O);
    bf16* Ot = reinterpret_cast<bf16*>(p.O);
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        float lt;
        if (FIX) lt = lsum[qf][0];
        else {
            lt = l[qf];
            lt += __shfl_xor(lt, 16, 64);
            lt += __shfl_xor(lt, 32, 64);
        }
        const float inv = 1.0f / lt;
        const int qi = q0 + qf * 16 + lr;
        if (qi < p.nq) {
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                const size_t off = ((size_t)s * p.nq + qi) * p.ldo + hh * DH + df * 16 + g * 4;
                const f32x4 v = o[qf][df] * inv;
                if (p.out_f32) store4(Of + off, v); else store4(Ot + off, v);
            }
            // training forward: the row's log-sum-exp for the backward kernels (fixed-offset form: p = 2^(s log2e - off2))
            if (p.lse && g == 0) p.lse[(size_t)sh * p.nq + qi] = FIX ? (p.off2 + __log2f(lt)) * 0.69314718055994530942f : m[qf] + __logf(lt);
        }
    }
}


// ---- small-sequence self-attention (n <= 64, no null keys): the C-ViViT spatial (n = 64) and temporal (n = 9..10)
// layers.  ONE launch replaces prep_q + prep_kv + attn_fwd: a wave owns G = 64 / n whole sequences of one head, stages
// the l2-normalised, k_scale'd keys and the values in LDS (f32), and every lane is one query row doing an f32 VALU
// flash loop over its sequence's keys (the work is ~0.3 GFLOP per video; what matters here is launch count and fixed
// cost, not MFMA).  Exact f32 arithmetic in both precision modes.
struct SmallAttnArgs {
    const float* q; int ldq;            // [S*n][ldq], head hh at columns hh*64
    const float* kv; int ldkv;          // [S*n][ldkv], k at [0, h*64), v at [h*64, 2*h*64)
    const float* q_scale; const float* k_scale; float scale;
    const float* bias; long bias_hstride; int bias_ld;
    const unsigned char* kmask;         // [S][n] or null
    const float* slopes; int causal;
    void* O; int ldo; int out_kind;     // 0: f32, 1: bf16
    int S, h, n;
};

constexpr int SROW = 68;                // LDS row stride in floats (64 + 4: rows land on different banks)

template <typename TO>
__global__ __launch_bounds__(64) void attn_small_kernel(const SmallAttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];          // K^ [G*n][SROW] | V [G*n][SROW]
    const int lane = threadIdx.x;
    const int G = 64 / p.n;
    const int groups = (p.S + G - 1) / G;
    const int grp = blockIdx.x % groups, hh = blockIdx.x / groups;
    const int s0 = grp * G;
    const int nseq = (p.S - s0) < G ? (p.S - s0) : G;
    const int rows = nseq * p.n;
    float* Ks = sm;
    float* Vs = sm + 64 * SROW;
    // stage keys / values: 16 lanes per row (one 16-byte load each), 4 rows per pass
    const int l16 = lane & 15;
    const f32x4 ks = *reinterpret_cast<const f32x4*>(p.k_scale + l16 * 4);
    for (int r = lane >> 4; r < ((rows + 3) & ~3); r += 4) {
        f32x4 kx = f32x4{0, 0, 0, 0}, vx = kx;
        if (r < rows) {
            const float* row = p.kv + ((size_t)s0 * p.n + r) * p.ldkv;
            kx = *reinterpret_cast<const f32x4*>(row + hh * DH + l16 * 4);
            vx = *reinterpret_cast<const f32x4*>(row + p.h * DH + hh * DH + l16 * 4);
        }
        const float ss = group16_sum((kx[0] * kx[0] + kx[1] * kx[1]) + (kx[2] * kx[2] + kx[3] * kx[3]));
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        if (r < rows) {
            f32x4 kn;
#pragma unroll
            for (int e = 0; e < 4; ++e) kn[e] = kx[e] * inv * ks[e];
            *reinterpret_cast<f32x4*>(Ks + r * SROW + l16 * 4) = kn;
            *reinterpret_cast<f32x4*>(Vs + r * SROW + l16 * 4) = vx;
        }
    }
    __syncthreads();
    if (lane >= rows) return;
    const int sl = lane / p.n, i = lane % p.n;                          // local sequence, query position
    const int s = s0 + sl;
    // q^ = l2norm(q) * q_scale * scale, kept in registers
    f32x4 qv[16];
    {
        const float* qrow = p.q + ((size_t)s * p.n + i) * p.ldq + hh * DH;
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            qv[c] = *reinterpret_cast<const f32x4*>(qrow + c * 4);
            ss += (qv[c][0] * qv[c][0] + qv[c][1] * qv[c][1]) + (qv[c][2] * qv[c][2] + qv[c][3] * qv[c][3]);
        }
        const float inv = p.scale / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 qs = *reinterpret_cast<const f32x4*>(p.q_scale + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[c][e] *= inv * qs[e];
        }
    }
    const float slope = (p.causal && p.slopes) ? p.slopes[hh] : 0.f;
    const float* brow = p.bias ? p.bias + (size_t)hh * p.bias_hstride + (size_t)i * p.bias_ld : nullptr;
    const unsigned char* km = p.kmask ? p.kmask + (size_t)s * p.n : nullptr;
    const float* Kq = Ks + sl * p.n * SROW;
    const float* Vq = Vs + sl * p.n * SROW;
    float m = -INFINITY, l = 0.f;
    f32x4 o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) o[c] = f32x4{0, 0, 0, 0};
    for (int j = 0; j < p.n; ++j) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 kk = *reinterpret_cast<const f32x4*>(Kq + j * SROW + c * 4);
            a0 += qv[c][0] * kk[0]; a1 += qv[c][1] * kk[1]; a2 += qv[c][2] * kk[2]; a3 += qv[c][3] * kk[3];
        }
        float sv = (a0 + a1) + (a2 + a3);
        if (brow) sv += brow[j];
        bool masked = km && !km[j];
        if (p.causal) {
            const int dj = j - i;
            sv -= fabsf((float)dj) * slope;
            masked = masked || dj > 0;
        }
        if (masked) sv = NEG_MAX;
        const float mn = fmaxf(m, sv);
        const float alpha = __expf(m - mn), pj = __expf(sv - mn);
        l = l * alpha + pj;
        m = mn;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(Vq + j * SROW + c * 4);
            o[c] = o[c] * alpha + vv * pj;
        }
    }
    const float inv = 1.0f / l;
    TO* orow = reinterpret_cast<TO*>(p.O) + ((size_t)s * p.n + i) * p.ldo + hh * DH;
#pragma unroll
    for (int c = 0; c < 16; ++c) store4(orow + c * 4, o[c] * inv);
}

}  // namespace pk
using namespace pk;

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// sizes (in elements of T) the caller must allocate: Qp = S*h*nq_pad*64, Kp = S*h*nk_pad*64, Vt = same as Kp
extern "C" int pk_attn_pads(int nq, int n_kv, int nnull, int* nq_pad, int* nk_pad) {
    if (nq <= 0 || n_kv <= 0 || nnull < 0 || !nq_pad || !nk_pad) return PK_EINVAL;
    *nq_pad = round_up(nq, nq >= 256 ? 64 : (nq >= 128 ? 32 : 16));
    *nk_pad = round_up(nnull + n_kv, 32);
    return PK_OK;
}

extern "C" int pk_attn_prep(int dtype, const float* q, int ldq, const float* kv, int ldkv, const float* null_kv,
                            const float* q_scale, const float* k_scale, float scale,
                            void* Qp, void* Kp, void* Vt, int S, int h, int nq, int n_kv, int nnull, void* stream) {
    if (!q || !Qp || S <= 0 || h <= 0) return PK_EINVAL;
    if (kv && (!Kp || !Vt)) return PK_EINVAL;                  // kv == NULL: query side only (cached K/V)
    if (kv && ((q_scale == nullptr) != (k_scale == nullptr))) return PK_EINVAL;      // both NULL: plain dot-product attention (no l2norm)
    if (kv && nnull > 0 && !null_kv) return PK_EINVAL;
    int nq_pad, nk_pad;
    if (int rc = pk_attn_pads(nq, n_kv, nnull, &nq_pad, &nk_pad)) return rc;
    auto mis = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) != 0; };
    if ((ldq & 3) || mis(q) || (q_scale && mis(q_scale)) || (kv && ((ldkv & 3) || mis(kv) || (k_scale && mis(k_scale)) || (nnull > 0 && mis(null_kv))))) return PK_EALIGN;
    PrepArgs p{q, ldq, kv, ldkv, null_kv, q_scale, k_scale, Qp, Kp, Vt, S, h, nq, n_kv, nnull, nq_pad, nk_pad, scale};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long qrows = (long)S * h * nq_pad;
    const int tiles = (nk_pad + 63) / 64;
    const unsigned nqb = (unsigned)((qrows + 15) / 16), nkb = kv ? (unsigned)(S * h * tiles) : 0u;
    if (dtype == 1) {
        hipLaunchKernelGGL((prep_qkv_kernel<bf16>), dim3(nqb + nkb), dim3(256), 0, s, p, (int)nqb);
    } else if (dtype == 0) {
        hipLaunchKernelGGL((prep_qkv_kernel<float>), dim3(nqb + nkb), dim3(256), 0, s, p, (int)nqb);
    } else if (dtype == 2) {
        // split-bf16: the images are pre-split (hi | lo) bf16 planes in 128-byte blocks of 32 elements (common.hpp bf16x3p); same sizes
        // as the f32 images; rows are 64 / nk_pad (% 32 == 0) elements, the bases must sit on a 128-byte boundary
        if ((reinterpret_cast<uintptr_t>(Qp) & 127) || (kv && ((reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vt)) & 127))) return PK_EALIGN;
        hipLaunchKernelGGL((prep_qkv_kernel<bf16x3p>), dim3(nqb + nkb), dim3(256), 0, s, p, (int)nqb);
    } else return PK_EINVAL;
    PK_CHECK_LAUNCH();
    return PK_OK;
}

#ifdef PK_TIMELINE
extern "C" int pk_debug_attn_timeline(unsigned long long* out, int n, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return PK_ELAUNCH;
    if (clear) { static unsigned long long z[8 * 6 * 16] = {}; return hipMemcpyToSymbol(HIP_SYMBOL(pk_attn_tl), z, sizeof(z)) == hipSuccess ? PK_OK : PK_ELAUNCH; }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pk_attn_tl), sizeof(unsigned long long) * n) == hipSuccess ? PK_OK : PK_ELAUNCH;
}
#endif

static int attn_fwd_impl(int dtype, const void* Qp, const void* Kp, const void* Vt,
                         const float* bias, long bias_hstride, int bias_ld, const unsigned char* kmask,
                         const float* slopes, int causal, void* O, int ldo, int out_is_f32,
                         int S, int h, int nq, int n_kv, int nnull, const float* bias_tab, int tab_len, const int* pos_code,
                         int code_off, int tab_run4, float score_bound, float* lse, void* stream) {
    if (!Qp || !Kp || !Vt || !O || S <= 0 || h <= 0) return PK_EINVAL;
    if (bias_tab && (bias || !pos_code || tab_len <= 0 || nnull != 0 || nq != n_kv || causal || kmask)) return PK_EINVAL;
    if (ldo & 3) return PK_EALIGN;
    int nq_pad, nk_pad;
    if (int rc = pk_attn_pads(nq, n_kv, nnull, &nq_pad, &nk_pad)) return rc;
    AttnArgs a{Qp, Kp, Vt, bias, bias_hstride, bias_ld, kmask, slopes, O, ldo, out_is_f32, S, h, nq, n_kv, nnull, nq_pad, nk_pad, causal, 0,
               bias_tab, tab_len, pos_code, code_off, tab_run4, -1.f, lse};
    a.bias_vec = (bias && nnull == 0 && (bias_ld & 3) == 0 && (bias_hstride & 3) == 0 &&
                  (reinterpret_cast<uintptr_t>(bias) & 15) == 0) ? 1 : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // query rows per wave: more rows amortise the K / V^T operand stream (the kernel is L1/L2 fetch bound); the
    // exact-f32 path stops at 32 rows (register budget)
    static const int qf_cap = [] { const char* e = getenv("PK_ATTN_MAX_QF"); return e ? atoi(e) : 2; }();   // tuning knob (4 measured slower: 146 vs 104 us)
    int QF = (nq >= 256 && dtype == 1) ? 4 : (nq >= 128 ? 2 : 1);
    if (QF > qf_cap && qf_cap >= 1) QF = qf_cap >= 2 ? 2 : 1;
    const long waves = (long)S * h * (nq_pad / (16 * QF));
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    static const int use_lds = [] { const char* e = getenv("PK_ATTN_LDS"); return e ? atoi(e) : 1; }();   // tuning knob
    if (dtype == 1 && use_lds && (!lse || (!bias_tab && !(score_bound == score_bound))) && nnull + n_kv >= 64 && nq >= 64 &&
        (size_t)S * h * nk_pad * 128 < 0xFFFFFFF0ull) {
        // measured on maskgit self-attention (S*h = 128, n = 576, bias): 16 query rows per wave + bias prefetch 41.2 us,
        // 32 rows per wave 44.5 us (its prefetch spills: 77 us); with a single key tile (n = 64) there is nothing to
        // prefetch and the leaner kernel (5 waves/SIMD) wins, 8.1 vs 11.3 us
        static const int lds_qf = [] { const char* e = getenv("PK_ATTN_LDS_QF"); return e ? atoi(e) : 0; }();   // tuning knobs (0: automatic)
        static const int pf_env = [] { const char* e = getenv("PK_ATTN_PF"); return e ? atoi(e) : -1; }();
        // fixed-offset softmax (below): 32 query rows per wave fit its register budget, and once 16-row waves would need more than one
        // round of workgroups (3 per CU with the bias table in LDS) the 128-row workgroups win: S*h = 128, n = 576: 26.6 vs 35.8 us
        // (table), 23.4 vs 29.9 (no bias); S*h = 96: 21.6 vs 30.1; at S*h <= 64 one round either way and 16 rows win (18.8 vs 20.5 us)
        const bool fix = score_bound == score_bound && fabsf(score_bound) < 1e4f && !kmask && !causal && !bias;
        static const int n_cu = [] { int dev = 0, cus = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256; return cus > 0 ? cus : 256; }();
        const long wgs16 = (long)S * h * ((nq_pad + 63) / 64);
        const int qf = ((lds_qf == 2 || (fix && lds_qf != 1 && wgs16 > 3L * n_cu)) && nq >= 128) ? 2 : 1;
        const bool pf = pf_env >= 0 ? pf_env != 0 : (qf == 1 && nk_pad >= 192);
        const int qblocks = (nq_pad + 64 * qf - 1) / (64 * qf);
        const uint32_t kv_bytes = (uint32_t)((size_t)S * h * nk_pad * 128);
        dim3 g2((unsigned)(S * h * qblocks));
        const size_t lds = (size_t)2 * 16384 + (bias_tab ? (((size_t)tab_len * 4 + 15) & ~(size_t)15) + (((size_t)n_kv * 4 + 15) & ~(size_t)15) : 0);
        if (lds > 65536) return PK_EINVAL;
        // fixed-offset softmax: a finite bound of sim + bias from the caller, no masks (a fully masked row needs the running maximum)
        if (fix) {
            a.off2 = ceilf(score_bound * ATTN_LOG2E);
            if (qf == 2) {
                if (bias_tab) hipLaunchKernelGGL((attn_fwd_lds_kernel<2, false, true, true>), g2, block, lds, s, a, kv_bytes);
                else hipLaunchKernelGGL((attn_fwd_lds_kernel<2, false, false, true>), g2, block, lds, s, a, kv_bytes);
            } else if (bias_tab) hipLaunchKernelGGL((attn_fwd_lds_kernel<1, false, true, true>), g2, block, lds, s, a, kv_bytes);
            else hipLaunchKernelGGL((attn_fwd_lds_kernel<1, false, false, true>), g2, block, lds, s, a, kv_bytes);
            PK_CHECK_LAUNCH();
            return PK_OK;
        }
        if (bias_tab) {
            if (qf == 2) hipLaunchKernelGGL((attn_fwd_lds_kernel<2, false, true>), g2, block, lds, s, a, kv_bytes);
            else hipLaunchKernelGGL((attn_fwd_lds_kernel<1, false, true>), g2, block, lds, s, a, kv_bytes);
            PK_CHECK_LAUNCH();
            return PK_OK;
        }
        if (qf == 2) {
            if (pf) hipLaunchKernelGGL((attn_fwd_lds_kernel<2, true>), g2, block, 32768, s, a, kv_bytes);
            else hipLaunchKernelGGL((attn_fwd_lds_kernel<2, false>), g2, block, 32768, s, a, kv_bytes);
        } else {
            if (pf) hipLaunchKernelGGL((attn_fwd_lds_kernel<1, true>), g2, block, 32768, s, a, kv_bytes);
            else hipLaunchKernelGGL((attn_fwd_lds_kernel<1, false>), g2, block, 32768, s, a, kv_bytes);
        }
        PK_CHECK_LAUNCH();
        return PK_OK;
    }
    if (dtype == 2 && use_lds && !lse && nnull + n_kv >= 64 && nq >= 128 && score_bound == score_bound && fabsf(score_bound) < 1e4f && !kmask && !causal && !bias &&
        out_is_f32 && (size_t)S * h * nk_pad * 256 < 0xFFFFFFF0ull &&
        !((reinterpret_cast<uintptr_t>(Qp) | reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vt)) & 127)) {
        // split-bf16 images, fixed-offset softmax (round 3): the same LDS-staged kernel on tiles twice as large (64 KB ring + the bias table:
        // one workgroup of 128 query rows per CU); 167 us for the LDS-free running-max kernel at S*h = 128, n = 576
        const int qblocks = (nq_pad + 127) / 128;
        const uint32_t kv_bytes = (uint32_t)((size_t)S * h * nk_pad * 256);
        const size_t lds = (size_t)2 * 32768 + (bias_tab ? (((size_t)tab_len * 4 + 15) & ~(size_t)15) + (((size_t)n_kv * 4 + 15) & ~(size_t)15) : 0);
        if (lds > 160 * 1024) return PK_EINVAL;
        static bool attr_tab = false, attr_plain = false;
        a.off2 = ceilf(score_bound * ATTN_LOG2E);
        dim3 g2((unsigned)(S * h * qblocks));
        if (bias_tab) {
            if (!attr_tab) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_lds_kernel<2, false, true, true, bf16x3p>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PK_ELAUNCH;
                attr_tab = true;
            }
            hipLaunchKernelGGL((attn_fwd_lds_kernel<2, false, true, true, bf16x3p>), g2, block, lds, s, a, kv_bytes);
        } else {
            if (!attr_plain) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_lds_kernel<2, false, false, true, bf16x3p>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PK_ELAUNCH;
                attr_plain = true;
            }
            hipLaunchKernelGGL((attn_fwd_lds_kernel<2, false, false, true, bf16x3p>), g2, block, lds, s, a, kv_bytes);
        }
        PK_CHECK_LAUNCH();
        return PK_OK;
    }
    if (dtype == 2 && lse && use_lds && nnull == 0 && nq == n_kv && nq >= 128 && !kmask && !causal && out_is_f32 && !bias_tab &&
        (!bias || a.bias_vec) && (size_t)S * h * nk_pad * 256 < 0xFFFFFFF0ull &&
        !((reinterpret_cast<uintptr_t>(Qp) | reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vt)) & 127)) {
        // training forward, split-bf16 (round 6): the LDS-staged kernel in its RUNNING-MAX form (the fixed-offset form needs a score bound on the
        // host, i.e. a read-back of the learnable scales / the bias table's extremes every step) with the position bias as a matrix; it also
        // writes lse.  The LDS-free kernel it replaces streams K / V^T through L1 per wave: 104-120 us at S h = 64, n = 576.
        static const bool on = !(getenv("PK_ATTN_TRAIN_LDS") && getenv("PK_ATTN_TRAIN_LDS")[0] == '0');      // A/B switch
        if (on) {
            static bool attr = false;
            if (!attr) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_lds_kernel<2, false, false, false, bf16x3p>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PK_ELAUNCH;
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_lds_kernel<3, false, false, false, bf16x3p>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PK_ELAUNCH;
                attr = true;
            }
            // one workgroup per CU (the 64 KB ring, one wave per SIMD): 128-row workgroups are 5 per head at n = 576, i.e. 320 for the 64 (sequence, head)
            // pairs of a B = 8 step = two rounds on 256 CUs; 192-row workgroups (48 rows per wave) are 192 = one round
            static const int n_cu = [] { int dev = 0, cus = 256; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256; return cus > 0 ? cus : 256; }();
            static const int qf_env = [] { const char* e = getenv("PK_ATTN_TRAIN_QF"); return e ? atoi(e) : 0; }();
            const long wg2 = (long)S * h * ((nq_pad + 127) / 128), wg3 = (long)S * h * ((nq_pad + 191) / 192);
            const bool three = qf_env ? qf_env == 3 : (wg2 > n_cu && wg3 <= n_cu);
            if (three) hipLaunchKernelGGL((attn_fwd_lds_kernel<3, false, false, false, bf16x3p>), dim3((unsigned)wg3), block, (size_t)2 * 32768, s, a,
                                          (uint32_t)((size_t)S * h * nk_pad * 256));
            else hipLaunchKernelGGL((attn_fwd_lds_kernel<2, false, false, false, bf16x3p>), dim3((unsigned)wg2), block, (size_t)2 * 32768, s, a,
                                    (uint32_t)((size_t)S * h * nk_pad * 256));
            PK_CHECK_LAUNCH();
            return PK_OK;
        }
    }
    if (bias_tab) return PK_EINVAL;                       // the table form exists in the LDS-staged kernel only
    if (dtype == 1) {
        if (QF == 4) hipLaunchKernelGGL((attn_fwd_kernel<bf16, 4>), grid, block, 0, s, a);
        else if (QF == 2) hipLaunchKernelGGL((attn_fwd_kernel<bf16, 2>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<bf16, 1>), grid, block, 0, s, a);
    } else if (dtype == 0) {
        if (QF == 2) hipLaunchKernelGGL((attn_fwd_kernel<float, 2>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<float, 1>), grid, block, 0, s, a);
    } else if (dtype == 2) {
        if (!out_is_f32 || ((reinterpret_cast<uintptr_t>(Qp) | reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vt)) & 127)) return PK_EINVAL;
        if (QF == 2) hipLaunchKernelGGL((attn_fwd_kernel<bf16x3p, 2>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attn_fwd_kernel<bf16x3p, 1>), grid, block, 0, s, a);
    } else return PK_EINVAL;
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_attn_fwd(int dtype, const void* Qp, const void* Kp, const void* Vt,
                           const float* bias, long bias_hstride, int bias_ld, const unsigned char* kmask,
                           const float* slopes, int causal, void* O, int ldo, int out_is_f32,
                           int S, int h, int nq, int n_kv, int nnull, const float* bias_tab, int tab_len, const int* pos_code,
                           int code_off, int tab_run4, float score_bound, void* stream) {
    return attn_fwd_impl(dtype, Qp, Kp, Vt, bias, bias_hstride, bias_ld, kmask, slopes, causal, O, ldo, out_is_f32, S, h, nq, n_kv, nnull, bias_tab, tab_len,
                         pos_code, code_off, tab_run4, score_bound, nullptr, stream);
}
// the training forward: the same product through the LDS-free kernel, which also writes lse (S h, nq) = the log-sum-exp of every score row --
// pk_attn_bwd (flags bit 1) then skips its own pass over the keys for it.  No bias table / fixed-offset form here.
extern "C" int pk_attn_fwd_lse(int dtype, const void* Qp, const void* Kp, const void* Vt,
                               const float* bias, long bias_hstride, int bias_ld, const unsigned char* kmask,
                               const float* slopes, int causal, void* O, int ldo, int out_is_f32,
                               int S, int h, int nq, int n_kv, int nnull, float* lse, void* stream) {
    if (!lse) return PK_EINVAL;
    return attn_fwd_impl(dtype, Qp, Kp, Vt, bias, bias_hstride, bias_ld, kmask, slopes, causal, O, ldo, out_is_f32, S, h, nq, n_kv, nnull, nullptr, 0,
                         nullptr, 0, 0, __builtin_nanf(""), lse, stream);
}

// attention.py:128-182 for short self-attention sequences (n <= 64, no null keys) straight from the projection outputs:
// one launch instead of pk_attn_prep + pk_attn_fwd; f32 arithmetic; O is f32 (out_kind 0) or bf16 (1)
extern "C" int pk_attn_small(const float* q, int ldq, const float* kv, int ldkv, const float* q_scale, const float* k_scale,
                             float scale, const float* bias, long bias_hstride, int bias_ld, const unsigned char* kmask,
                             const float* slopes, int causal, void* O, int ldo, int out_kind, int S, int h, int n, void* stream) {
    if (!q || !kv || !q_scale || !k_scale || !O || S <= 0 || h <= 0 || n <= 0 || n > 64) return PK_EINVAL;
    auto mis = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) != 0; };
    if ((ldq & 3) || (ldkv & 3) || (ldo & 3) || mis(q) || mis(kv) || mis(q_scale) || mis(k_scale) || mis(O)) return PK_EALIGN;
    SmallAttnArgs a{q, ldq, kv, ldkv, q_scale, k_scale, scale, bias, bias_hstride, bias_ld, kmask, slopes, causal, O, ldo, out_kind, S, h, n};
    const int G = 64 / n, groups = (S + G - 1) / G;
    const size_t lds = (size_t)2 * 64 * SROW * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (out_kind == 0) hipLaunchKernelGGL((attn_small_kernel<float>), dim3((unsigned)(groups * h)), dim3(64), lds, s, a);
    else hipLaunchKernelGGL((attn_small_kernel<bf16>), dim3((unsigned)(groups * h)), dim3(64), lds, s, a);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
