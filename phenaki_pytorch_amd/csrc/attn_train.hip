// Attention backward for the training step (reference attention.py:132-182 under autograd).
//   q^ = l2norm(q) * q_scale * scale,  k^ = l2norm([null_k ; k]) * k_scale,  v_all = [null_v ; v]
//   S = q^ k^T + bias (+ key mask),  P = softmax(S),  O = P v_all
// The forward pass of a training step runs the inference kernels (pk_attn_prep / pk_attn_fwd); what is saved is q, kv (the projection
// outputs), O and dO.  Backward recomputes the f32 operands (pk_attn_train_prep), then
//   pk_attn_bwd :  kernel Q (one workgroup per 64 query rows of a head): lse by an online pass over the keys, D = rowsum(dO * O),
//                  dS = P * (dO V^T - D), dQ^ = dS K^   [+ dS written out for the position-bias gradient]
//                  kernel KV (one workgroup per 64 keys of a head): dV = P^T dO, dK^ = dS^T Q^ over all query tiles
//   pk_attn_train_prep_bwd : l2norm / scale / null-key backward -> dq, dkv, dq_scale, dk_scale, dnull_kv
// All tile products run on v_mfma_f32_16x16x4_f32 from 8-element fragment chunks; a workgroup's own rows (the A side) stay in registers, B
// operands are shared through f32 LDS tiles [row][k] (rows padded to 68 / 36 floats); products whose contraction index is not the contiguous one
// of the global layout read a tile that was transposed on its way into LDS (K^T in kernel Q, q^T / dO^T in kernel KV); the next tile is
// fetched into registers while the current one is multiplied.  Every result is owned by exactly one workgroup (no atomics): the gradients
// are bit-reproducible.
#include <cstdlib>
#include "common.hpp"

namespace pk {

#define STREAM(s) reinterpret_cast<hipStream_t>(s)
constexpr int TLD = 68;                        // LDS tile row stride (floats)
constexpr int TSZ = 64 * TLD;                  // one 64 x 64 tile

// ---- f32 operand images of one attention call: Qh [S*h][n][64], Kh / Vh [S*h][nkt][64], nkt = nnull + n_kv --------------------------
// 16 lanes per row, 4 consecutive head dims (one 16-byte access) per lane, the row's sums by a 4-step xor shuffle inside the 16-lane group -- the
// arrangement (and summation order) of the forward's prep_q / prep_kv kernels (round 6: one 4-byte access per lane and a 64-lane reduction per row
// before: 18 / 24 us for 56 / 85 MB)
__device__ __forceinline__ float g16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}
__device__ __forceinline__ float sumsq4(const f32x4& x) { return (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]); }
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }

__global__ __launch_bounds__(256) void attn_train_prep_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ kv, long ldkv,
                                                              const float* __restrict__ null_kv, const float* __restrict__ q_scale,
                                                              const float* __restrict__ k_scale, float scale, float* __restrict__ Qh,
                                                              float* __restrict__ Kh, float* __restrict__ Vh, int S, int heads, int n, int n_kv,
                                                              int nnull, long tasks) {
    const int l16 = threadIdx.x & 15, c = l16 * 4;
    const int nkt = nnull + n_kv, inner = heads * 64;
    const long nq = (long)S * heads * n;
    const f32x4 qs = *reinterpret_cast<const f32x4*>(q_scale + c), ks = *reinterpret_cast<const f32x4*>(k_scale + c);
    for (long t = (long)blockIdx.x * 16 + (threadIdx.x >> 4); t < tasks; t += (long)gridDim.x * 16) {
        if (t < nq) {
            const int i = (int)(t % n), h = (int)((t / n) % heads), s = (int)(t / ((long)n * heads));
            const f32x4 x = *reinterpret_cast<const f32x4*>(q + ((long)s * n + i) * ldq + h * 64 + c);
            const float inv = scale / fmaxf(sqrtf(g16_sum(sumsq4(x))), 1e-12f);
            *reinterpret_cast<f32x4*>(Qh + t * 64 + c) = x * inv * qs;
        } else {
            const long u = t - nq;
            const int j = (int)(u % nkt), h = (int)((u / nkt) % heads), s = (int)(u / ((long)nkt * heads));
            f32x4 k, v;
            if (j < nnull) {
                k = *reinterpret_cast<const f32x4*>(null_kv + ((long)h * 2 * nnull + 2 * j) * 64 + c);
                v = *reinterpret_cast<const f32x4*>(null_kv + ((long)h * 2 * nnull + 2 * j + 1) * 64 + c);
            } else {
                const float* row = kv + ((long)s * n_kv + (j - nnull)) * ldkv + h * 64 + c;
                k = *reinterpret_cast<const f32x4*>(row);
                v = *reinterpret_cast<const f32x4*>(row + inner);
            }
            const float inv = 1.0f / fmaxf(sqrtf(g16_sum(sumsq4(k))), 1e-12f);
            *reinterpret_cast<f32x4*>(Kh + u * 64 + c) = k * inv * ks;
            *reinterpret_cast<f32x4*>(Vh + u * 64 + c) = v;
        }
    }
}

// backward of the above.  dq (M, ldq'), dkv (Mk, ldkv') are overwritten for the real rows; pq / pk: (gridDim.x, 64) partials of dq_scale / dk_scale
__global__ __launch_bounds__(256) void attn_train_prep_bwd_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ kv, long ldkv,
                                                                  const float* __restrict__ null_kv, const float* __restrict__ q_scale,
                                                                  const float* __restrict__ k_scale, float scale, const float* __restrict__ dQh,
                                                                  const float* __restrict__ dKh, const float* __restrict__ dVh, float* __restrict__ dq,
                                                                  long lddq, float* __restrict__ dkv, long lddkv, float* __restrict__ pq,
                                                                  float* __restrict__ pk_, int S, int heads, int n, int n_kv, int nnull, long tasks) {
    __shared__ f32x4 red[2][16][16];
    const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4, c = l16 * 4;
    const int nkt = nnull + n_kv, inner = heads * 64;
    const long nq = (long)S * heads * n;
    const f32x4 qs = *reinterpret_cast<const f32x4*>(q_scale + c), ks = *reinterpret_cast<const f32x4*>(k_scale + c);
    f32x4 aq = {0.f, 0.f, 0.f, 0.f}, ak = aq;
    for (long t = (long)blockIdx.x * 16 + grp; t < tasks; t += (long)gridDim.x * 16) {
        if (t < nq) {
            const int i = (int)(t % n), h = (int)((t / n) % heads), s = (int)(t / ((long)n * heads));
            const f32x4 x = *reinterpret_cast<const f32x4*>(q + ((long)s * n + i) * ldq + h * 64 + c);
            const float inv = 1.0f / fmaxf(sqrtf(g16_sum(sumsq4(x))), 1e-12f);
            const f32x4 xn = x * inv, g = *reinterpret_cast<const f32x4*>(dQh + t * 64 + c);
            aq += g * xn * scale;
            const f32x4 dn = g * qs * scale;
            const float dot = g16_sum(dot4(dn, xn));
            *reinterpret_cast<f32x4*>(dq + ((long)s * n + i) * lddq + h * 64 + c) = (dn - xn * dot) * inv;
        } else {
            const long u = t - nq;
            const int j = (int)(u % nkt), h = (int)((u / nkt) % heads), s = (int)(u / ((long)nkt * heads));
            const f32x4 k = j < nnull ? *reinterpret_cast<const f32x4*>(null_kv + ((long)h * 2 * nnull + 2 * j) * 64 + c)
                                      : *reinterpret_cast<const f32x4*>(kv + ((long)s * n_kv + (j - nnull)) * ldkv + h * 64 + c);
            const float inv = 1.0f / fmaxf(sqrtf(g16_sum(sumsq4(k))), 1e-12f);
            const f32x4 kn = k * inv, g = *reinterpret_cast<const f32x4*>(dKh + u * 64 + c);
            ak += g * kn;
            if (j >= nnull) {
                const f32x4 dn = g * ks;
                const float dot = g16_sum(dot4(dn, kn));
                float* row = dkv + ((long)s * n_kv + (j - nnull)) * lddkv + h * 64 + c;
                *reinterpret_cast<f32x4*>(row) = (dn - kn * dot) * inv;
                *reinterpret_cast<f32x4*>(row + inner) = *reinterpret_cast<const f32x4*>(dVh + u * 64 + c);
            }
        }
    }
    red[0][grp][l16] = aq;
    red[1][grp][l16] = ak;
    __syncthreads();
    if (threadIdx.x < 32) {                                              // thread (which = tid >> 4, column lane l16): the 16 row groups in order
        const int which = threadIdx.x >> 4;
        f32x4 sum = red[which][0][l16];
#pragma unroll
        for (int g2 = 1; g2 < 16; ++g2) sum += red[which][g2][l16];
        float* dst = which == 0 ? pq : pk_;
        const long pst = (pk_ == pq + 64) ? 128 : 64;                    // the two halves of one (parts, 128) buffer: one pk_colsum finishes both
        *reinterpret_cast<f32x4*>(dst + (long)blockIdx.x * pst + c) = sum;
    }
}
// dnull_kv[h][2 j + {0, 1}][d] = sum over the sequences of the null key's l2norm backward / of dV at the null slot (one wave per (h, j))
__global__ __launch_bounds__(64) void null_kv_bwd_kernel(const float* __restrict__ null_kv, const float* __restrict__ k_scale, const float* __restrict__ dKh,
                                                         const float* __restrict__ dVh, float* __restrict__ dnull, int S, int heads, int nkt, int nnull) {
    const int lane = threadIdx.x, j = blockIdx.x % nnull, h = blockIdx.x / nnull;
    const float k = null_kv[((long)h * 2 * nnull + 2 * j) * 64 + lane];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(k * k)), 1e-12f);
    const float kn = k * inv, ks = k_scale[lane];
    float gk = 0.f, gv = 0.f;
    for (int s = 0; s < S; ++s) {
        const long u = ((long)s * heads + h) * nkt + j;
        gk += dKh[u * 64 + lane];
        gv += dVh[u * 64 + lane];
    }
    const float dn = gk * ks;                                   // the l2norm backward is linear in the incoming gradient: sum first
    const float dot = wave_sum(dn * kn);
    dnull[((long)h * 2 * nnull + 2 * j) * 64 + lane] = (dn - kn * dot) * inv;
    dnull[((long)h * 2 * nnull + 2 * j + 1) * 64 + lane] = gv;
}

// ---- tile products on the f32 matrix cores ------------------------------------------------------------------------------------------------
// Operands are f32 [row][k] with k contiguous, read as 8-element fragment chunks (lane: row = lane & 15, k = (lane >> 4) * 8 + 0..7; two 16-byte
// reads feed eight v_mfma_f32_16x16x4_f32, common.hpp mma(Frag<float>)).  The A side of every product is either held in REGISTERS for the whole
// kernel (the workgroup's own q^ / dO rows in kernel Q, its k^ / v rows in kernel KV) or a wave-private LDS tile (P / dS, written in the
// accumulator layout and read back as fragments); only B operands are shared through LDS.  acc[nb]: rows (lane >> 4) * 4 + i, column nb * 16 + (lane & 15).
// X3 = split-bf16 tile products (the bf16x3 / bf16 compute modes): the f32 fragments are split into (hi, lo) bf16 planes in registers
// (common.hpp split8, 24 VALU per 8 values) and every 16 x 16 x 32 block costs three v_mfma_f32_16x16x32_bf16 (3 x 16 cycles) instead of eight
// v_mfma_f32_16x16x4_f32 (8 x 32 cycles).  Same-box A/B: training step 37.3 -> 36.0 ms (bf16x3), 31.7 -> 30.5 ms (bf16).  Keeping the tiles in
// LDS as ready-made (hi | lo) bf16 planes (split once per workgroup when stashed, no conversions in the inner loops) was built and measured
// too: no further gain (the loops wait on LDS / barriers, not on the VALU splits), so the simpler f32 tiles stay.
// ~2^-17 relative per product, as everywhere else in the bf16x3 mode.  X3 = false: exact f32 (the 'fp32' mode).
// X3 = 2 (round 6, the bf16 compute mode only): ONE bf16 MFMA per block on the RNE-rounded fragments -- the products the bf16 forward kernels make of
// the same q^ / k^ / v, so the forward's log-sum-exp is the backward's too; a third of the MFMAs and 4 instead of 24 conversion instructions per fragment.
template <int X3> struct OpFragOf { typedef Frag<float> type; };
template <> struct OpFragOf<1> { typedef Frag<bf16x3p> type; };
template <> struct OpFragOf<2> { typedef Frag<bf16> type; };
__device__ __forceinline__ void to_operand(Frag<float>& o, const Frag<float>& f) { o = f; }
__device__ __forceinline__ void to_operand(Frag<bf16x3p>& o, const Frag<float>& f) { split8(f.lo, f.hi, o.hi, o.lo); }
__device__ __forceinline__ void to_operand(Frag<bf16>& o, const Frag<float>& f) {
    o.v = u32x4{pack_bf2(f.lo[0], f.lo[1]), pack_bf2(f.lo[2], f.lo[3]), pack_bf2(f.hi[0], f.hi[1]), pack_bf2(f.hi[2], f.hi[3])};
}

template <int NBLK, int NCHUNK, typename OF>
__device__ __forceinline__ void mma_regA(const OF (&a)[NCHUNK], const float* Bs, int ldb, f32x4 (&acc)[NBLK], int lane) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            Frag<float> bf;
            frag_load(bf, Bs + (nb * 16 + r) * ldb + c * 32 + kq * 8);
            OF b;
            to_operand(b, bf);
            acc[nb] = mma(a[c], b, acc[nb]);
        }
}
template <int NBLK, int NCHUNK, typename OF>
__device__ __forceinline__ void mma_ldsA(const float* As, int lda, int m0, const float* Bs, int ldb, f32x4 (&acc)[NBLK], int lane) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        Frag<float> af;
        frag_load(af, As + (m0 + r) * lda + c * 32 + kq * 8);
        OF a;
        to_operand(a, af);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            Frag<float> bf;
            frag_load(bf, Bs + (nb * 16 + r) * ldb + c * 32 + kq * 8);
            OF b;
            to_operand(b, bf);
            acc[nb] = mma(a, b, acc[nb]);
        }
    }
}

struct AttnBwdArgs {
    const float* Qh; const float* Kh; const float* Vh;       // [S*h][n][64], [S*h][nkt][64]
    const void* O; long ldo_; int o_bf16;                    // forward output (M, >= heads * 64), f32 or bf16
    const float* dO; long lddo;                              // (M, >= heads * 64) f32
    const float* bias;                                       // [heads][n][nkt - nnull] or null (columns = real keys)
    const unsigned char* kmask;                              // [S][nkt - nnull] or null
    const float* slopes; int causal;                         // causal: ALiBi -|j - (nkt - n + i)| slopes[h] over ALL keys + the causal mask
    float* dQh; float* dKh; float* dVh;
    float* dS;                                               // [S*h][n][nkt - nnull] or null (real-key columns only)
    float* lse; float* Drow;                                 // [S*h][n] scratch written by kernel Q, read by kernel KV
    int S, heads, n, nkt, nnull;
    // PACKED short sequences (pack_n > 0; self-attention without null keys, pack_n <= 32): one workgroup tile holds pack_g = 64 / pack_n whole
    // (sequence, head) groups -- the fields above then describe VIRTUAL heads (S = number of tiles, heads = 1, n = nkt = pack_g * pack_n rows of the
    // flat [S*h][n][64] arrays) and scores exist only inside a group (block-diagonal); act_heads / act_groups = the real heads / S * heads.
    int pack_n, pack_g, act_heads, act_groups, pack_inv;     // pack_inv = 65536 / pack_n + 1: row / pack_n = (row * pack_inv) >> 16 for row < 64
    int have_lse;                                            // lse was written by the forward (pk_attn_fwd_lse): kernel Q skips its own pass over the keys
    // kernel KV with few key tiles (cross-attention: 14 keys = ONE tile per head, 64 workgroups walking 18 query tiles each, 75 us): the query tiles
    // are dealt to kv_split workgroups per key tile, each writing its partial dK^ / dV to kv_part (+ kv_split slabs of the dKh layout, dV behind dK);
    // the host adds the slabs in index order (deterministic).  kv_split <= 1: one workgroup per key tile writes dKh / dVh itself.
    int kv_split, kv_chunk; float* kv_part;
};

// element offset of row vr (of this workgroup's virtual head sh) in the heads-merged O / dO matrices, or -1 beyond the data
__device__ __forceinline__ long merged_row_offset(const AttnBwdArgs& p, int sh, int vr, long ld) {
    if (p.pack_n == 0) {
        if (vr >= p.n) return -1;
        const int h = sh % p.heads, s = sh / p.heads;
        return ((long)s * p.n + vr) * ld + h * 64;
    }
    const int g = (vr * p.pack_inv) >> 16;
    const int group = sh * p.pack_g + g;
    if (g >= p.pack_g || group >= p.act_groups) return -1;
    return ((long)(group / p.act_heads) * p.pack_n + (vr - g * p.pack_n)) * ld + (group % p.act_heads) * 64;
}
// rows of the flat per-head arrays this workgroup may touch
__device__ __forceinline__ int live_rows(const AttnBwdArgs& p, int sh) {
    if (p.pack_n == 0) return p.n;
    const long left = (long)p.act_groups * p.pack_n - (long)sh * p.n;
    return left < p.n ? (int)left : p.n;
}

// this wave's 16 rows [row0, row0 + 16) x 64 columns of a (rows_total, ld) matrix as two fragment chunks (zero rows beyond rows_total)
__device__ __forceinline__ void load_rows_frag(Frag<float> (&f)[2], const float* src, long ld, int row0, int rows_total, int lane) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (row0 + r < rows_total) frag_load(f[c], src + (long)(row0 + r) * ld + c * 32 + kq * 8);
        else frag_zero(f[c]);
    }
}
// the same from the heads-merged O / dO matrices (row addresses through merged_row_offset)
__device__ __forceinline__ void load_rows_frag_merged(Frag<float> (&f)[2], const float* src, long ld, const AttnBwdArgs& p, int sh, int row0, int lane) {
    const int r = lane & 15, kq = lane >> 4;
    const long off = merged_row_offset(p, sh, row0 + r, ld);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (off >= 0) frag_load(f[c], src + off + c * 32 + kq * 8);
        else frag_zero(f[c]);
    }
}
template <int ROWS>
__device__ __forceinline__ void fetch_tile_merged(f32x4 (&v)[ROWS / 16], const float* src, long ld, const AttnBwdArgs& p, int sh, int r0) {
    const int rr = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int u = 0; u < ROWS / 16; ++u) {
        const long off = merged_row_offset(p, sh, r0 + rr + 16 * u, ld);
        v[u] = off >= 0 ? *reinterpret_cast<const f32x4*>(src + off + c) : f32x4{0, 0, 0, 0};
    }
}
// a thread's share of a [ROWS x 64] tile of a (rows_total, ld) matrix: ROWS / 16 pieces of 4 floats (piece u: row (tid >> 4) + 16 u, columns (tid & 15) * 4 ..)
template <int ROWS>
__device__ __forceinline__ void fetch_tile(f32x4 (&v)[ROWS / 16], const float* src, long ld, int r0, int rows_total) {
    const int rr = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int u = 0; u < ROWS / 16; ++u) {
        const int row = r0 + rr + 16 * u;
        v[u] = row < rows_total ? *reinterpret_cast<const f32x4*>(src + (long)row * ld + c) : f32x4{0, 0, 0, 0};
    }
}
// ... into LDS as [row][col] (stride ld_n) and / or transposed [col][row] (stride ld_t)
template <int ROWS>
__device__ __forceinline__ void stash_tile(const f32x4 (&v)[ROWS / 16], float* normal, int ld_n, float* transposed, int ld_t) {
    const int rr = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
#pragma unroll
    for (int u = 0; u < ROWS / 16; ++u) {
        const int row = rr + 16 * u;
        if (normal) *reinterpret_cast<f32x4*>(normal + row * ld_n + c) = v[u];
        if (transposed) {
#pragma unroll
            for (int q = 0; q < 4; ++q) transposed[(c + q) * ld_t + row] = v[u][q];
        }
    }
}

// score of (query row gi, key j) from the raw product, or "excluded"
__device__ __forceinline__ bool score(const AttnBwdArgs& p, int s, int h, int gi, int j, float raw, float& out) {
    if (j >= p.nkt || gi >= p.n) return false;
    if (p.pack_n) {                                            // s = the virtual head (tile) index: (gi, j) interact only inside one real group
        const int g = (gi * p.pack_inv) >> 16;
        if (((j * p.pack_inv) >> 16) != g) return false;
        const int group = s * p.pack_g + g;
        if (g >= p.pack_g || group >= p.act_groups) return false;
        const int li = gi - g * p.pack_n, lj = j - g * p.pack_n;
        float v = raw;
        if (p.bias) v += p.bias[((long)(group % p.act_heads) * p.pack_n + li) * p.pack_n + lj];
        if (p.kmask && !p.kmask[(long)(group / p.act_heads) * p.pack_n + lj]) v = NEG_MAX;
        if (p.causal) {
            const int d = lj - li;
            v = d > 0 ? NEG_MAX : v + (float)d * p.slopes[group % p.act_heads];
        }
        out = v;
        return true;
    }
    const int jr = j - p.nnull;                                // real-key column
    float v = raw;
    if (p.bias && jr >= 0) v += p.bias[((long)h * p.n + gi) * (p.nkt - p.nnull) + jr];
    if (p.kmask && jr >= 0 && !p.kmask[(long)s * (p.nkt - p.nnull) + jr]) v = NEG_MAX;
    if (p.causal) {                                            // attention.py:166-172: the null keys sit in front and are never masked
        const int d = j - (p.nkt - p.n + gi);
        v = d > 0 ? NEG_MAX : v + (float)d * p.slopes[h];     // d <= 0: -|d| slope = d slope
    }
    out = v;
    return true;
}

#define PK_ZERO4 {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}
#define PK_ZERO2 {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}}

// kernel Q: one workgroup per 64 query rows of a head (4 waves x 16 rows).  q^ and dO fragments live in registers; LDS holds K^ [64][68],
// K^T [64][68], V [64][68] of the current 64-key tile and the wave-private dS rows (70 KB: two workgroups per CU).  The next tile's K^ / V rows
// are fetched into registers while the current one is being multiplied.
// 2 waves per SIMD (two workgroups per CU, which the 70 KB of LDS allow): without the bound the compiler took 255 VGPRs + 42 AGPRs = ONE wave per
// SIMD -- 256 resident workgroups, so the 576 of a B = 8 step ran in three rounds
// (round 6, measured and removed: a variant whose dS rows shared the V tile's LDS -- 52 KB and a 168-register cap, three workgroups per CU so that the 576
// workgroups of a B = 8 step run in one round instead of 512 + 64 -- spilled 210 VGPRs and lost: training step 30.13 vs 29.34 ms same-box.)
template <int X3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_q_kernel(const AttnBwdArgs p) {
    typedef typename OpFragOf<X3>::type OF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ks = sm; float* Vs = sm + TSZ; float* Kt = sm + 2 * TSZ; float* Ps = sm + 3 * TSZ;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
    const int nqt = (p.n + 63) / 64;
    const int qt = blockIdx.x % nqt, sh = blockIdx.x / nqt, h = sh % p.heads, s = sh / p.heads;
    const int i0 = qt * 64, m0 = wv * 16;
    const int nreal = p.nkt - p.nnull;
    const int nlive = live_rows(p, sh);
    Frag<float> fq[2], fdo[2];
    load_rows_frag(fq, p.Qh + (long)sh * p.n * 64, 64, i0 + m0, nlive, lane);
    load_rows_frag_merged(fdo, p.dO, p.lddo, p, sh, i0 + m0, lane);
    OF fqx[2], fdox[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) { to_operand(fqx[c], fq[c]); to_operand(fdox[c], fdo[c]); }
    // D = rowsum(dO * O) for this wave's 16 rows, from the dO fragments already in registers and the matching O fragments: a lane multiplies
    // its 16 elements of row (lane & 15), the four k-quarters are folded with two shuffles, and the accumulator layout's rows (kq * 4 + i) are
    // picked from the lanes that hold them (before: 16 dependent rounds of two global loads + a 64-lane reduction -- 70 of the kernel's 107 us
    // at n = 64)
    float Dl[4];
    {
        const long oo = merged_row_offset(p, sh, i0 + m0 + r, p.ldo_);
        float part = 0.f;
        if (oo >= 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x4 olo, ohi;
                if (p.o_bf16) {
                    const u32x4 raw = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(p.O) + oo + c * 32 + kq * 8);
                    olo = f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u), __uint_as_float(raw[1] << 16), __uint_as_float(raw[1] & 0xffff0000u)};
                    ohi = f32x4{__uint_as_float(raw[2] << 16), __uint_as_float(raw[2] & 0xffff0000u), __uint_as_float(raw[3] << 16), __uint_as_float(raw[3] & 0xffff0000u)};
                } else {
                    const float* op = reinterpret_cast<const float*>(p.O) + oo + c * 32 + kq * 8;
                    olo = *reinterpret_cast<const f32x4*>(op);
                    ohi = *reinterpret_cast<const f32x4*>(op + 4);
                }
                const f32x4 t = olo * fdo[c].lo + ohi * fdo[c].hi;
                part += (t[0] + t[1]) + (t[2] + t[3]);
            }
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);                         // every lane of row r now holds D[row r]
        if (kq == 0 && i0 + m0 + r < nlive) p.Drow[(long)sh * p.n + i0 + m0 + r] = part;
#pragma unroll
        for (int i = 0; i < 4; ++i) Dl[i] = __shfl(part, kq * 4 + i, 64);
    }
    const int ntl = (p.nkt + 63) / 64;
    const float* Kbase = p.Kh + (long)sh * p.nkt * 64;
    const float* Vbase = p.Vh + (long)sh * p.nkt * 64;
    // ---- pass 1: log-sum-exp of every row (online, per lane over its 4 rows x 16 columns per tile; lanes of a row merged at the end) -- unless the
    // forward kernel handed it over (round 6, have_lse: a quarter of this kernel's products at n = 576)
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, l[4] = {0.f, 0.f, 0.f, 0.f};
    const int klive = p.pack_n ? nlive : p.nkt;
    f32x4 kreg[4], vreg[4];
    float lse[4];
    if (p.have_lse) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gi = i0 + m0 + kq * 4 + i;
            lse[i] = gi < nlive ? p.lse[(long)sh * p.n + gi] : 0.f;
        }
    } else {
    fetch_tile<64>(kreg, Kbase, 64, 0, klive);
    for (int kt = 0; kt < ntl; ++kt) {
        __syncthreads();                                          // the previous tile's readers are done
        stash_tile<64>(kreg, Ks, TLD, nullptr, 0);
        if (kt + 1 < ntl) fetch_tile<64>(kreg, Kbase, 64, (kt + 1) * 64, klive);
        __syncthreads();
        f32x4 acc[4] = PK_ZERO4;
        mma_regA<4, 2, OF>(fqx, Ks, TLD, acc, lane);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sc;
                if (!score(p, s, h, i0 + m0 + kq * 4 + i, kt * 64 + nb * 16 + r, acc[nb][i], sc)) continue;
                const float mn = fmaxf(mx[i], sc);
                l[i] = l[i] * __expf(mx[i] - mn) + __expf(sc - mn);
                mx[i] = mn;
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const float m2 = __shfl_xor(mx[i], off, 64), l2 = __shfl_xor(l[i], off, 64);
            const float mn = fmaxf(mx[i], m2);
            const float a = mx[i] == -INFINITY ? 0.f : l[i] * __expf(mx[i] - mn), b = m2 == -INFINITY ? 0.f : l2 * __expf(m2 - mn);
            l[i] = a + b;
            mx[i] = mn;
        }
        lse[i] = mx[i] + __logf(l[i]);
        const int gi = i0 + m0 + kq * 4 + i;
        if (r == 0 && gi < nlive) p.lse[(long)sh * p.n + gi] = lse[i];
    }
    }
    // ---- pass 2: dS and dQ^
    f32x4 accQ[4] = PK_ZERO4;
    fetch_tile<64>(kreg, Kbase, 64, 0, klive);
    fetch_tile<64>(vreg, Vbase, 64, 0, klive);
    for (int kt = 0; kt < ntl; ++kt) {
        __syncthreads();
        stash_tile<64>(kreg, Ks, TLD, Kt, TLD);
        stash_tile<64>(vreg, Vs, TLD, nullptr, 0);
        if (kt + 1 < ntl) {
            fetch_tile<64>(kreg, Kbase, 64, (kt + 1) * 64, klive);
            fetch_tile<64>(vreg, Vbase, 64, (kt + 1) * 64, klive);
        }
        __syncthreads();
        f32x4 accS[4] = PK_ZERO4;
        f32x4 accP[4] = PK_ZERO4;
        mma_regA<4, 2, OF>(fqx, Ks, TLD, accS, lane);
        mma_regA<4, 2, OF>(fdox, Vs, TLD, accP, lane);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gi = i0 + m0 + kq * 4 + i, j = kt * 64 + nb * 16 + r;
                float sc, ds = 0.f;
                if (score(p, s, h, gi, j, accS[nb][i], sc)) ds = __expf(sc - lse[i]) * (accP[nb][i] - Dl[i]);
                Ps[(m0 + kq * 4 + i) * TLD + nb * 16 + r] = ds;
            }
        if (p.dS) {
            // the score gradient leaves through the wave's own LDS rows (round 6): 16 lanes x 16 bytes = one whole 256-byte row segment per store
            // instead of 16 four-byte stores per lane in the accumulator layout (85 MB per layer at n = 576: kernel Q 190 vs 115 us without dS)
            const int col = (lane & 15) * 4, j = kt * 64 + col, jr = j - p.nnull;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = m0 + u * 4 + (lane >> 4), gi = i0 + row;
                if (gi >= p.n) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(Ps + row * TLD + col);
                float* dst = p.dS + ((long)sh * p.n + gi) * nreal + jr;
                if (jr >= 0 && j + 3 < p.nkt && !(nreal & 3) && !(jr & 3)) *reinterpret_cast<f32x4*>(dst) = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (jr + e >= 0 && j + e < p.nkt) dst[e] = v[e];
                }
            }
        }
        // this wave's 16 rows of Ps are its own: no workgroup barrier before reading them back.  dQ^[i][d] += sum_j dS[i][j] K^T[d][j]
        mma_ldsA<4, 2, OF>(Ps, TLD, m0, Kt, TLD, accQ, lane);
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gi = i0 + m0 + kq * 4 + i;
            if (gi < nlive) p.dQh[((long)sh * p.n + gi) * 64 + nb * 16 + r] = accQ[nb][i];
        }
}

// kernel KV: one workgroup per 64 keys of a head (4 waves x 16 keys); k^ and v fragments live in registers.  Query rows come in tiles of 32:
// LDS holds q^ [32][68], dO [32][68] (B operands of the TRANSPOSED scores S^T[j][i], dP^T[j][i]), their transposes [64][36] (B operands of
// dV = P^T dO, dK^ = dS^T q^) and one wave-private [64][36] buffer that carries P^T and then dS^T (46 KB: three workgroups per CU).
constexpr int QT = 32, TLQ = 36;
template <int X3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bwd_kv_kernel(const AttnBwdArgs p) {
    typedef typename OpFragOf<X3>::type OF;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Qs = sm; float* dOs = sm + QT * TLD; float* Qt = sm + 2 * QT * TLD; float* dOt = Qt + 64 * TLQ; float* PS = dOt + 64 * TLQ;
    float* lse_s = PS + 64 * TLQ; float* D_s = lse_s + QT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
    const int ntl = (p.nkt + 63) / 64;
    const int G = p.kv_split > 1 ? p.kv_split : 1;
    const int gq = blockIdx.x % G, tile = blockIdx.x / G;
    const int kt = tile % ntl, sh = tile / ntl, h = sh % p.heads, s = sh / p.heads;
    const int j0 = kt * 64, m0 = wv * 16;
    const int nlive = live_rows(p, sh), klive = p.pack_n ? nlive : p.nkt;
    Frag<float> fk[2], fv[2];
    load_rows_frag(fk, p.Kh + (long)sh * p.nkt * 64, 64, j0 + m0, klive, lane);
    load_rows_frag(fv, p.Vh + (long)sh * p.nkt * 64, 64, j0 + m0, klive, lane);
    OF fkx[2], fvx[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) { to_operand(fkx[c], fk[c]); to_operand(fvx[c], fv[c]); }
    f32x4 accK[4] = PK_ZERO4;
    f32x4 accV[4] = PK_ZERO4;
    const int nqt_all = (p.n + QT - 1) / QT;
    const int qt0 = G > 1 ? gq * p.kv_chunk : 0;
    const int nqt = G > 1 ? (qt0 + p.kv_chunk < nqt_all ? qt0 + p.kv_chunk : nqt_all) : nqt_all;
    const float* Qbase = p.Qh + (long)sh * p.n * 64;
    f32x4 qreg[2], dreg[2];
    fetch_tile<QT>(qreg, Qbase, 64, qt0 * QT, nlive);
    fetch_tile_merged<QT>(dreg, p.dO, p.lddo, p, sh, qt0 * QT);
    for (int qt = qt0; qt < nqt; ++qt) {
        const int i0 = qt * QT;
        __syncthreads();                                          // the previous tile's readers are done
        stash_tile<QT>(qreg, Qs, TLD, Qt, TLQ);
        stash_tile<QT>(dreg, dOs, TLD, dOt, TLQ);
        if (threadIdx.x < QT) {
            const int gi = i0 + threadIdx.x;
            lse_s[threadIdx.x] = gi < nlive ? p.lse[(long)sh * p.n + gi] : 0.f;
            D_s[threadIdx.x] = gi < nlive ? p.Drow[(long)sh * p.n + gi] : 0.f;
        }
        if (qt + 1 < nqt) {
            fetch_tile<QT>(qreg, Qbase, 64, i0 + QT, nlive);
            fetch_tile_merged<QT>(dreg, p.dO, p.lddo, p, sh, i0 + QT);
        }
        __syncthreads();
        f32x4 accS[2] = PK_ZERO2;
        f32x4 accP[2] = PK_ZERO2;
        mma_regA<2, 2, OF>(fkx, Qs, TLD, accS, lane);             // S^T[j][i]
        mma_regA<2, 2, OF>(fvx, dOs, TLD, accP, lane);                 // dP^T[j][i] = sum_d V[j][d] dO[i][d]
        float pr[2][4], dsv[2][4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int ci = nb * 16 + r, gi = i0 + ci;
            const float lse = lse_s[ci], Dr = D_s[ci];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float sc;
                pr[nb][i] = 0.f;
                if (score(p, s, h, gi, j0 + m0 + kq * 4 + i, accS[nb][i], sc)) pr[nb][i] = __expf(sc - lse);
                dsv[nb][i] = pr[nb][i] * (accP[nb][i] - Dr);
                PS[(m0 + kq * 4 + i) * TLQ + ci] = pr[nb][i];
            }
        }
        // dV[j][d] += sum_i P^T[j][i] dO^T[d][i]   (A rows = this wave's own 16 rows of PS: program order is enough)
        mma_ldsA<4, 1, OF>(PS, TLQ, m0, dOt, TLQ, accV, lane);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) PS[(m0 + kq * 4 + i) * TLQ + nb * 16 + r] = dsv[nb][i];
        // dK^[j][d] += sum_i dS^T[j][i] q^T[d][i]
        mma_ldsA<4, 1, OF>(PS, TLQ, m0, Qt, TLQ, accK, lane);
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gj = j0 + m0 + kq * 4 + i;
            if (gj < klive) {
                const long o = ((long)sh * p.nkt + gj) * 64 + nb * 16 + r;
                if (G > 1) {
                    const long slab = (long)p.S * p.heads * p.nkt * 64;
                    p.kv_part[(long)gq * slab + o] = accK[nb][i];
                    p.kv_part[(long)(G + gq) * slab + o] = accV[nb][i];
                } else {
                    p.dKh[o] = accK[nb][i];
                    p.dVh[o] = accV[nb][i];
                }
            }
        }
}

}  // namespace pk

using namespace pk;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static constexpr int KV_SMEM = (2 * QT * TLD + 3 * 64 * TLQ + 2 * QT) * 4;

extern "C" int pk_attn_train_prep(const float* q, long ldq, const float* kv, long ldkv, const float* null_kv, const float* q_scale, const float* k_scale,
                                  float scale, float* Qh, float* Kh, float* Vh, int S, int heads, int n, int n_kv, int nnull, void* stream) {
    if (!q || !kv || !q_scale || !k_scale || !Qh || !Kh || !Vh || S <= 0 || heads <= 0 || n <= 0 || n_kv <= 0 || nnull < 0 || (nnull > 0 && !null_kv)) return PK_EINVAL;
    if ((ldq & 3) || (ldkv & 3) || !al16(q) || !al16(kv) || !al16(q_scale) || !al16(k_scale) || !al16(Qh) || !al16(Kh) || !al16(Vh) || (nnull > 0 && !al16(null_kv))) return PK_EALIGN;
    const long tasks = (long)S * heads * (n + nnull + n_kv);
    long blocks = (tasks + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(attn_train_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM(stream), q, ldq, kv, ldkv, null_kv, q_scale, k_scale, scale, Qh, Kh, Vh,
                       S, heads, n, n_kv, nnull, tasks);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// pq / pk: (PK_ATTN_PREP_BWD_PARTS = 1024, 64) partials of dq_scale / dk_scale (finish with pk_colsum); dnull (heads, 2 nnull, 64) or null
extern "C" int pk_attn_train_prep_bwd(const float* q, long ldq, const float* kv, long ldkv, const float* null_kv, const float* q_scale, const float* k_scale,
                                      float scale, const float* dQh, const float* dKh, const float* dVh, float* dq, long lddq, float* dkv, long lddkv,
                                      float* pq, float* pk_, float* dnull, int S, int heads, int n, int n_kv, int nnull, void* stream) {
    if (!q || !kv || !q_scale || !k_scale || !dQh || !dKh || !dVh || !dq || !dkv || !pq || !pk_ || S <= 0 || heads <= 0 || n <= 0 || n_kv <= 0 || nnull < 0 ||
        (nnull > 0 && (!null_kv || !dnull))) return PK_EINVAL;
    if ((ldq & 3) || (ldkv & 3) || (lddq & 3) || (lddkv & 3) || !al16(q) || !al16(kv) || !al16(q_scale) || !al16(k_scale) || !al16(dQh) || !al16(dKh) || !al16(dVh) ||
        !al16(dq) || !al16(dkv) || !al16(pq) || !al16(pk_) || (nnull > 0 && !al16(null_kv))) return PK_EALIGN;
    const long tasks = (long)S * heads * (n + nnull + n_kv);
    hipStream_t s = STREAM(stream);
    hipLaunchKernelGGL(attn_train_prep_bwd_kernel, dim3(1024), dim3(256), 0, s, q, ldq, kv, ldkv, null_kv, q_scale, k_scale, scale, dQh, dKh, dVh,
                       dq, lddq, dkv, lddkv, pq, pk_, S, heads, n, n_kv, nnull, tasks);
    if (nnull > 0)
        hipLaunchKernelGGL(null_kv_bwd_kernel, dim3(heads * nnull), dim3(64), 0, s, null_kv, k_scale, dKh, dVh, dnull, S, heads, nnull + n_kv, nnull);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// O: the forward output (M = S n rows, ldo elements per row; o_bf16 = 1: bf16), dO its gradient (f32).  bias (heads, n, n_kv) / kmask (S, n_kv)
// cover the REAL keys (the nnull leading null keys carry no bias and are never masked, attention.py:151-158).  dS (S heads, n, n_kv) optional.
// causal (self-attention, attention.py:166-172): ALiBi with slopes [heads] over all nnull + n keys, then key j > nnull + i masked.
// lse / Drow: (S heads n) f32 scratch.  split_bf16 is a flag word: bit 0 = split-bf16 tile products, bit 1 = lse already holds the log-sum-exp of
// every score row (written by pk_attn_fwd_lse in the forward pass): kernel Q then skips its own pass over the keys.
extern "C" int pk_sum_batch_multi(const void* jobs, int count, void* stream);
namespace { struct SumJobHost { const float* src; float* out; long stride, E4; int S, blk0; }; }   // = PkSumJob (include/phenaki_hip.h)

// floats of workspace pk_attn_bwd_ws wants for the shape (0: the query tiles are not dealt out)
static int kv_split_of(int S, int heads, int n, int nkt, int* chunk) {
    const int ntl = (nkt + 63) / 64, nqt = (n + QT - 1) / QT;
    const long wgs = (long)S * heads * ntl;
    int G = (int)(768 / (wgs > 0 ? wgs : 1));
    if (G < 2 || nqt < 2) { *chunk = nqt; return 1; }
    if (G > nqt) G = nqt;
    *chunk = (nqt + G - 1) / G;
    return (nqt + *chunk - 1) / *chunk;
}
extern "C" int pk_attn_bwd_work(int S, int heads, int n, int n_kv, int nnull) {
    if (S <= 0 || heads <= 0 || n <= 0 || n_kv <= 0 || nnull < 0) return PK_EINVAL;
    int chunk;
    const int G = kv_split_of(S, heads, n, nnull + n_kv, &chunk);
    return G > 1 ? (int)(2L * G * S * heads * (nnull + n_kv) * 64) : 0;          // G > 1 only below 384 key tiles: < 2^24 floats
}
extern "C" int pk_attn_bwd_ws(const float* Qh, const float* Kh, const float* Vh, const void* O, long ldo, int o_bf16, const float* dO, long lddo,
                              const float* bias, const unsigned char* kmask, const float* slopes, int causal, float* dQh, float* dKh, float* dVh, float* dS,
                              float* lse, float* Drow, int S, int heads, int n, int n_kv, int nnull, int split_bf16, float* work, long work_floats, void* stream);
extern "C" int pk_attn_bwd(const float* Qh, const float* Kh, const float* Vh, const void* O, long ldo, int o_bf16, const float* dO, long lddo,
                           const float* bias, const unsigned char* kmask, const float* slopes, int causal, float* dQh, float* dKh, float* dVh, float* dS,
                           float* lse, float* Drow, int S, int heads, int n, int n_kv, int nnull, int split_bf16, void* stream) {
    return pk_attn_bwd_ws(Qh, Kh, Vh, O, ldo, o_bf16, dO, lddo, bias, kmask, slopes, causal, dQh, dKh, dVh, dS, lse, Drow, S, heads, n, n_kv, nnull, split_bf16,
                          nullptr, 0, stream);
}
// work (pk_attn_bwd_work floats, or NULL): lets kernel KV deal the query tiles of a key tile to several workgroups when there are few key tiles
extern "C" int pk_attn_bwd_ws(const float* Qh, const float* Kh, const float* Vh, const void* O, long ldo, int o_bf16, const float* dO, long lddo,
                              const float* bias, const unsigned char* kmask, const float* slopes, int causal, float* dQh, float* dKh, float* dVh, float* dS,
                              float* lse, float* Drow, int S, int heads, int n, int n_kv, int nnull, int split_bf16, float* work, long work_floats, void* stream) {
    if (!Qh || !Kh || !Vh || !O || !dO || !dQh || !dKh || !dVh || !lse || !Drow || S <= 0 || heads <= 0 || n <= 0 || n_kv <= 0 || nnull < 0) return PK_EINVAL;
    if (!al16(Qh) || !al16(Kh) || !al16(Vh) || !al16(dO) || (lddo & 3) || !al16(O) || (ldo & (o_bf16 ? 7 : 3))) return PK_EALIGN;
    if ((long)S * heads > 0x7fffffffL / 64) return PK_EINVAL;
    if (causal && (!slopes || n != n_kv)) return PK_EINVAL;
    const int have_lse = (split_bf16 >> 1) & 1;                         // flags: bit 0 split-bf16 tile products, bit 1 lse (S heads, n) given by pk_attn_fwd_lse,
    const bool bf16_products = (split_bf16 >> 2) & 1;                   // bit 2: single bf16 products (the bf16 compute mode)
    split_bf16 &= 1;
    AttnBwdArgs p{Qh, Kh, Vh, O, ldo, o_bf16, dO, lddo, bias, kmask, slopes, causal, dQh, dKh, dVh, dS, lse, Drow, S, heads, n, nnull + n_kv, nnull, 0, 0, heads, S * heads, 0, have_lse};
    static const bool pack_on = !(getenv("PK_ATTN_BWD_PACK") && getenv("PK_ATTN_BWD_PACK")[0] == '0');      // A/B switch
    // (the packed layout re-indexes the rows: lse is laid out [S heads][n], the flat index of a packed row is the same (S h n contiguous) -> usable as is)
    if (pack_on && nnull == 0 && n == n_kv && n <= 32 && !dS) {
        // short self-attention (the C-ViViT temporal transformers: n = 9 at 512 sequences x 8 heads): 64 / n whole (sequence, head) groups per tile
        // instead of one -- the flat [S*h][n][64] arrays are the same memory either way
        p.pack_n = n;
        p.pack_inv = 65536 / n + 1;
        p.pack_g = 64 / n;
        p.n = p.nkt = p.pack_g * n;
        p.heads = 1;
        p.S = (int)((p.act_groups + p.pack_g - 1) / p.pack_g);
    }
    hipStream_t s = STREAM(stream);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TSZ * 4) != hipSuccess) return PK_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TSZ * 4) != hipSuccess) return PK_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TSZ * 4) != hipSuccess) return PK_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kv_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, KV_SMEM) != hipSuccess) return PK_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kv_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, KV_SMEM) != hipSuccess) return PK_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kv_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, KV_SMEM) != hipSuccess) return PK_ELAUNCH;
        attr_done = true;
    }
    const int nqt = (p.n + 63) / 64, nktt = (p.nkt + 63) / 64;
    if (work && !p.pack_n) {
        int chunk;
        const int G = kv_split_of(p.S, p.heads, p.n, p.nkt, &chunk);
        if (G > 1) {
            if (work_floats < 2L * G * p.S * p.heads * p.nkt * 64 || !al16(work)) return PK_EINVAL;
            p.kv_split = G; p.kv_chunk = chunk; p.kv_part = work;
        }
    }
    const dim3 gq((unsigned)((long)p.S * p.heads * nqt)), gk((unsigned)((long)p.S * p.heads * nktt * (p.kv_split > 1 ? p.kv_split : 1)));
    static const bool split_on = !(getenv("PK_ATTN_BWD_SPLIT") && getenv("PK_ATTN_BWD_SPLIT")[0] == '0');     // A/B switch
    if (bf16_products) {
        hipLaunchKernelGGL(attn_bwd_q_kernel<2>, gq, dim3(256), 4 * TSZ * 4, s, p);
        hipLaunchKernelGGL(attn_bwd_kv_kernel<2>, gk, dim3(256), KV_SMEM, s, p);
    } else if (split_bf16 && split_on) {
        hipLaunchKernelGGL(attn_bwd_q_kernel<1>, gq, dim3(256), 4 * TSZ * 4, s, p);
        hipLaunchKernelGGL(attn_bwd_kv_kernel<1>, gk, dim3(256), KV_SMEM, s, p);
    } else {
        hipLaunchKernelGGL(attn_bwd_q_kernel<0>, gq, dim3(256), 4 * TSZ * 4, s, p);
        hipLaunchKernelGGL(attn_bwd_kv_kernel<0>, gk, dim3(256), KV_SMEM, s, p);
    }
    PK_CHECK_LAUNCH();
    if (p.kv_split > 1) {                                                // dK^ / dV = the slabs added in index order (one launch for both)
        const long slab = (long)p.S * p.heads * p.nkt * 64;
        SumJobHost jobs[2] = {{work, dKh, slab, slab / 4, p.kv_split, 0}, {work + (long)p.kv_split * slab, dVh, slab, slab / 4, p.kv_split, 0}};
        return pk_sum_batch_multi(jobs, 2, stream);
    }
    return PK_OK;
}
