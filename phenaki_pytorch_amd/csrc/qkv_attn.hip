// pk_qkv_attn (bf16 / split-bf16): a whole short-sequence self-attention block of the C-ViViT transformers (reference attention.py:142-182;
// the spatial layers n = 64 with the continuous position bias, cvivit.py:460-466, and the causal temporal layers n = 9..10 with
// ALiBi, cvivit.py:468-472) in ONE launch per layer:
//   to_q / to_kv projections (MFMA, LDS-DMA ring)  ->  l2norm, q_scale / k_scale, similarity scale  ->  softmax(q k^T + bias) v
// A workgroup owns one head of R = floor(64 / n) * n consecutive rows = floor(64 / n) WHOLE sequences (n = 64: one sequence, n = 9:
// seven), so every key a query needs is produced by the same workgroup: q^, k^ and v go to LDS as the bf16 operand images the
// attention MFMAs read (the same images, rounding points and fragment reads as pk_qkv_project + pk_attn_fwd, which this replaces for
// n <= 64: Qp / Kp / Vt never reach HBM -- the V^T scatter alone made the n = 9 projection 20 us -- and one launch per layer goes).
// Sequences sharing a tile are kept apart by a block-diagonal mask (other sequences' keys get weight exp(-inf) = 0 exactly).
// Two GEMM passes over the rows (q: 64 x 64 x K; k|v: 64 x 128 x K) on one 2-stage ring whose dead stages also stage q^ / k^ / v^T:
// 48 KB, 3 workgroups per CU.
// Roofline: MFMA (2 * R * 192 * K flops per workgroup) -- in practice bound by the L1 -> LDS fill rate like every short-K GEMM here.
// T = bf16x3 (round 4, the parity-grade mode): x rows f32 (split in registers by the main loop), weights host-split; q^ / k^ / v^T are staged
// as TWO bf16 planes (hi image + lo image 8 KB apart, the same slot swizzle in both), every attention product is three MFMAs (mma_split),
// P is split in registers, O is written f32.  Same 48 KB of LDS.  Replaces that mode's LayerNorm + to_q + to_kv + pk_attn_prep + pk_attn_fwd.
#include "gemm_dma.hpp"

namespace pk {

struct QkvAttnArgs {
    const void* xq; const void* xkv;        // [M][ld] bf16: LayerNorm(x) rows (queries) and the un-normalised x rows (keys / values)
    const void* wq; const void* wkv;        // [h*64][ldw], [2*h*64][ldw] bf16, K zero-padded to the k-tile
    int ld, ldw;
    int S, n, h, K;                         // S sequences of n <= 64 tokens, M = S * n
    int spt;                                // sequences per tile = 64 / n
    const float* q_scale; const float* k_scale; float scale;
    const float* bias; long bias_hstride; int bias_ld;     // [h][n][n] f32 or null
    const float* slopes; int causal;        // ALiBi slopes [h] with causal
    void* O; int ldo;                       // bf16 [(s*n + i)][hh*64 + d]
    uint32_t recip;                         // ceil(65536 / n): x / n == (x * recip) >> 16 for x < 64
    const float* q_ln_s;                    // LayerNorm folded into to_q (see pk_qkv_project): xq = the un-normalised rows, wq = gamma (.) Wq
};

template <typename T> using QaTile = GemmDma<T, 1, 4, 4, 1, 2, 128>;          // 64 rows x 64 columns (one head), 4 waves stacked on the rows
template <typename T> using QaKvTile = GemmDma<T, 1, 8, 4, 1, 2, 128>;        // 64 rows x 128 columns: the head's k | v columns in ONE pass over x
constexpr int QA_SMEM = QaKvTile<bf16>::SMEM;             // 48 KB (either type): the ring of the k|v pass; q^ / k^ / v^T staging aliases it
static_assert(QaKvTile<bf16x3>::SMEM == QA_SMEM, "the split-bf16 ring has the same bytes");

// staged images: one [64 rows][128 B] bf16 plane (bf16) or two of them 8 KB apart (split-bf16: hi, lo)
constexpr int QA_PLANE = 8192;
template <typename T> struct QaImg { static constexpr int PLANES = 1; };
template <> struct QaImg<bf16x3> { static constexpr int PLANES = 2; };
// 4 consecutive elements of a staged row: `off` = row * 128 + (slot << 4) + half * 8
__device__ __forceinline__ void qa_put4(bf16, char* img, int off, f32x4 v) {
    *reinterpret_cast<u32x2*>(img + off) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}
__device__ __forceinline__ void qa_put4(bf16x3, char* img, int off, f32x4 v) {
    const uint32_t h0 = pack_bf2(v[0], v[1]), h1 = pack_bf2(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h0 << 16), r1 = v[1] - __builtin_bit_cast(float, h0 & 0xFFFF0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h1 << 16), r3 = v[3] - __builtin_bit_cast(float, h1 & 0xFFFF0000u);
    *reinterpret_cast<u32x2*>(img + off) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(img + QA_PLANE + off) = u32x2{pack_bf2(r0, r1), pack_bf2(r2, r3)};
}
__device__ __forceinline__ void qa_put1(bf16, char* img, int off, float v) { *reinterpret_cast<u16*>(img + off) = f2bf(v); }
__device__ __forceinline__ void qa_put1(bf16x3, char* img, int off, float v) {
    const u16 h = f2bf(v);
    *reinterpret_cast<u16*>(img + off) = h;
    *reinterpret_cast<u16*>(img + QA_PLANE + off) = f2bf(v - bf2f(h));
}
// one fragment chunk (8 elements, one 16-byte slot) of a staged row
__device__ __forceinline__ void qa_get(Frag<bf16>& f, const char* img, int off) { f.v = *reinterpret_cast<const u32x4*>(img + off); }
__device__ __forceinline__ void qa_get(Frag<bf16x3>& f, const char* img, int off) {
    f.hi = *reinterpret_cast<const u32x4*>(img + off);
    f.lo = *reinterpret_cast<const u32x4*>(img + QA_PLANE + off);
}

// K tile rows permuted / swizzled exactly like pk_attn_fwd's LDS kernel (attn.hip): see attn_kperm / attn_ksw there
__device__ __forceinline__ int qa_kperm(int f, int i) { return (f >> 1) * 32 + (i >> 2) * 8 + (f & 1) * 4 + (i & 3); }
__device__ __forceinline__ int qa_ksw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

// q^ rows of one 64-row tile of head hh -> Qs [row][64] bf16 (16-B slot ^ (row & 7)); p.A / p.W / p.N / strides set by the caller
template <typename T>
__device__ __forceinline__ void qa_project_q(const GemmOperands& p, int M, int m0, int hh, int K, const float* q_scale, float scale,
                                             const float* q_ln_s, char* smem, char* Qs) {
    const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
    const int rq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 16 + lr;
    f32x4 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[0][j] = f32x4{0, 0, 0, 0};
    if (q_ln_s) {                                                   // LayerNorm folded into to_q; the l2norm cancels rstd: row mean and s suffice
        float rsum[1], rsq[1];
        (void)QaTile<T>::template run_stats<1>(p, M, m0, hh * 64, smem, acc, rsum, rsq);
        const float mean = rsum[0] / (float)K;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(q_ln_s + hh * 64 + j * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][j][r] -= mean * s4[r];
        }
    } else {
        (void)QaTile<T>::run(p, M, m0, hh * 64, smem, acc);
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ss += acc[0][j][r] * acc[0][j][r];
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float inv = scale / fmaxf(sqrtf(ss), 1e-12f);              // F.normalize eps = 1e-12
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(q_scale + j * 16 + g * 4);
        f32x4 v = acc[0][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= inv * sc[r];
        const int slot = (2 * j + (g >> 1)) ^ (rq & 7);
        qa_put4(T{}, Qs, rq * 128 + (slot << 4) + (g & 1) * 8, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void qkv_attn_kernel(const QkvAttnArgs a) {
    // LDS: ONE 48 KB ring.  q pass (32 KB of it) -> q^ staged in the dead ring -> each lane pulls its two q^ fragments into 8 VGPRs ->
    // k|v pass (48 KB) -> k^ and v^T staged in the dead ring -> attention.  48 KB per workgroup: 3 workgroups per CU.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = QaImg<T>::PLANES;
    char* Qs = smem;
    char* Ks = smem;
    char* Vts = smem + NPL * QA_PLANE;
    // block -> (row tile, head): the 8 heads of a row tile run on ONE XCD (block b is observed on XCD b % 8; speed only), so the
    // two A tiles they all read are fetched over the fabric once
    const int R = a.spt * a.n;
    const int tiles = (a.S + a.spt - 1) / a.spt;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = (idx / a.h) * 8 + xcd, hh = idx % a.h;
    if (tile >= tiles) return;
    const int M = a.S * a.n;
    const int m0 = tile * R;
    const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rq = wave * 16 + lr;                                     // this lane's row of the tile (query / key index)

    GemmOperands p;
    p.a_rows = nullptr;
    p.lda = a.ld; p.ldw = a.ldw;
    p.M = M; p.K = a.K;
    p.plain_map = 0; p.krot = 0;
    p.w_gap_from = 0; p.w_gap_rows = 0;

    // ---- q^ = l2norm(LN(x) Wq^T) * q_scale * scale  -> Qs [row][64] bf16 (16-B slot ^ (row & 7)) -> this lane's fragments
    p.A = a.xq; p.W = a.wq; p.N = a.h * 64;
    qa_project_q<T>(p, M, m0, hh, a.K, a.q_scale, a.scale, a.q_ln_s, smem, Qs);
    Frag<T> fq[2];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // a wave reads only rows its own lanes wrote
#pragma unroll
    for (int c = 0; c < 2; ++c) qa_get(fq[c], Qs, rq * 128 + (((c * 4 + g) ^ (rq & 7)) << 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                      // every wave holds its q^ : the ring may be refilled

    // ---- k | v = x [Wk ; Wv]^T of head hh in one pass (tile columns 0..63 = k, 64..127 = v: wkv rows hh*64 + c and (h + hh)*64 + c - 64)
    {
        f32x4 acc[1][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[0][j] = f32x4{0, 0, 0, 0};
        p.A = a.xkv; p.W = a.wkv; p.N = 2 * a.h * 64;
        p.w_gap_from = 64; p.w_gap_rows = (a.h - 1) * 64;
        (void)QaKvTile<T>::run(p, M, m0, hh * 64, smem, acc);           // ends with a barrier: the ring is dead
        // k^ = l2norm(k) * k_scale -> Ks [key][64] bf16, slot ^ qa_ksw(key)
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ss += acc[0][j][r] * acc[0][j][r];
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.k_scale + j * 16 + g * 4);
            f32x4 v = acc[0][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= inv * sc[r];
            const int slot = (2 * j + (g >> 1)) ^ qa_ksw(rq);
            qa_put4(T{}, Ks, rq * 128 + (slot << 4) + (g & 1) * 8, v);
        }
        // v -> V^T [dim][64 keys] bf16, slot ^ (dim & 7)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = j * 16 + g * 4 + r;
                qa_put1(T{}, Vts, d * 128 + ((((rq >> 3) ^ (d & 7))) << 4) + (rq & 7) * 2, acc[0][4 + j][r]);
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- attention of this wave's 16 query rows against the 64 keys of the tile (block-diagonal over the tile's sequences)
    f32x4 st[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) st[f] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int krow = qa_kperm(f, lr);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Frag<T> fk;
            qa_get(fk, Ks, krow * 128 + (((c * 4 + g) ^ qa_ksw(krow)) << 4));
            st[f] = mma(fk, fq[c], st[f]);
        }
    }
    const int seq_q = (int)(((uint32_t)rq * a.recip) >> 16), pos_q = rq - seq_q * a.n;
    const bool qvalid = rq < R && tile * a.spt + seq_q < a.S;
    const float slope = (a.causal && a.slopes) ? a.slopes[hh] : 0.f;
    const float* brow = a.bias ? a.bias + (size_t)hh * a.bias_hstride + (size_t)pos_q * a.bias_ld : nullptr;
    const bool whole = a.n == 64 && !a.causal;                  // one sequence per tile, every key valid: no per-element masks
    float pr[16];
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int k0 = qa_kperm(f, g * 4);                      // the lane's 4 keys of block f are k0 .. k0 + 3
        f32x4 bz = f32x4{0, 0, 0, 0};
        if (whole && brow && (a.bias_ld & 3) == 0) bz = *reinterpret_cast<const f32x4*>(brow + k0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sv = st[f][r];
            if (whole) {
                if (brow) sv += (a.bias_ld & 3) == 0 ? bz[r] : brow[k0 + r];
            } else {
                const int kk = k0 + r;
                const int seq_k = (int)(((uint32_t)kk * a.recip) >> 16), pos_k = kk - seq_k * a.n;
                if (kk >= R || seq_k != seq_q) sv = -INFINITY;                 // not a key of this query's sequence: weight 0
                else {
                    if (brow) sv += brow[pos_k];
                    if (a.causal) {
                        const int dj = pos_k - pos_q;
                        sv -= fabsf((float)dj) * slope;                        // ALiBi, attention.py:198-227 (i == j: no offset)
                        if (dj > 0) sv = NEG_MAX;                              // causal mask fill, attention.py:172-174
                    }
                }
            }
            pr[f * 4 + r] = sv;
            mx = fmaxf(mx, sv);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (!qvalid) mx = 0.f;                                      // rows past the tile's sequences: keep the arithmetic finite, never stored
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { pr[e] = __expf(pr[e] - mx); ls += pr[e]; }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    f32x4 o[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) o[df] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        Frag<T> fp;
        {
            const float p8[8] = {pr[kc * 8 + 0], pr[kc * 8 + 1], pr[kc * 8 + 2], pr[kc * 8 + 3], pr[kc * 8 + 4], pr[kc * 8 + 5], pr[kc * 8 + 6], pr[kc * 8 + 7]};
            frag_from_f32(fp, p8);
        }
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const int d = df * 16 + lr;
            Frag<T> fv;
            qa_get(fv, Vts, d * 128 + (((kc * 4 + g) ^ (d & 7)) << 4));
            o[df] = mma(fv, fp, o[df]);
        }
    }
    if (!qvalid) return;
    const float inv = 1.0f / ls;
    T* orow = reinterpret_cast<T*>(a.O) + (size_t)(m0 + rq) * a.ldo + hh * 64;
#pragma unroll
    for (int df = 0; df < 4; ++df) store4(orow + df * 16 + g * 4, o[df] * inv);
}

// ---- cross-attention against CACHED keys / values (the step-invariant text context of MaskGit / TokenCritic, attention.py:142-182
// with context; <= 64 keys incl. the null keys): to_q (+ folded LayerNorm) + l2norm + softmax(q k^T) v in ONE launch.  Replaces
// pk_qkv_project (query side) + pk_attn_fwd: Qp (9.4 MB at 2 x 8 x 576 rows) is neither written nor read, one launch per layer and
// step goes.  A workgroup = 64 consecutive rows of one sequence (n % 64 == 0) x one head; K^ / V^T of that (sequence, head) come from
// the images pk_attn_prep left ([S][h][nk_pad][64], [S][h][64][nk_pad]).
struct QAttnCachedArgs {
    const void* xq; const void* wq; int ld, ldw;
    int S, n, h, K;
    const float* q_scale; float scale; const float* q_ln_s;
    const void* Kp; const void* Vt; int nk_pad, nk, nnull;
    const unsigned char* kmask; int n_kv;     // [S][n_kv] over the real keys (1 = keep) or null
    void* O; int ldo;
};

template <typename T, bool NK64>
__global__ __launch_bounds__(256) void q_attn_cached_kernel(const QAttnCachedArgs a) {
    // [ring 32 KB | Qs | Ks]  (Qs / Ks: 8 KB each for bf16, 16 KB = two planes each for split-bf16); V^T reuses the dead ring
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPL = QaImg<T>::PLANES;
    char* Qs = smem + QaTile<T>::SMEM;
    char* Ks = Qs + NPL * QA_PLANE;
    char* Vts = smem;
    const int M = a.S * a.n;
    const int tiles = M / 64;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = (idx / a.h) * 8 + xcd, hh = idx % a.h;
    if (tile >= tiles) return;
    const int m0 = tile * 64;
    const int s = m0 / a.n;
    const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rq = wave * 16 + lr;

    GemmOperands p;
    p.a_rows = nullptr;
    p.lda = a.ld; p.ldw = a.ldw;
    p.M = M; p.K = a.K;
    p.plain_map = 0; p.krot = 0;
    p.w_gap_from = 0; p.w_gap_rows = 0;
    p.A = a.xq; p.W = a.wq; p.N = a.h * 64;
    qa_project_q<T>(p, M, m0, hh, a.K, a.q_scale, a.scale, a.q_ln_s, smem, Qs);   // its main loop ends with a barrier: the ring is dead

    // K^ [key][64] -> Ks (slot ^ qa_ksw(key)), V^T [dim][keys] -> Vts (64-key rows of 128 B, slot ^ (dim & 7)); keys >= nk_pad: zeros
    {
        const size_t sh = (size_t)s * a.h + hh;
        const int row = threadIdx.x >> 2, part = threadIdx.x & 3;          // 64 rows x 4 parts
        const u32x4 zero = u32x4{0, 0, 0, 0};
        if constexpr (NPL == 1) {
            const bf16* krow = reinterpret_cast<const bf16*>(a.Kp) + (sh * a.nk_pad + row) * 64;
            const bf16* vrow = reinterpret_cast<const bf16*>(a.Vt) + (sh * 64 + row) * a.nk_pad;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int slot = part * 2 + q;
                const u32x4 kv = row < a.nk_pad ? *reinterpret_cast<const u32x4*>(krow + slot * 8) : zero;
                *reinterpret_cast<u32x4*>(Ks + row * 128 + ((slot ^ qa_ksw(row)) << 4)) = kv;
                const u32x4 vv = slot * 8 < a.nk_pad ? *reinterpret_cast<const u32x4*>(vrow + slot * 8) : zero;     // nk_pad % 32 == 0
                *reinterpret_cast<u32x4*>(Vts + row * 128 + ((slot ^ (row & 7)) << 4)) = vv;
            }
        } else {
            // pre-split images (common.hpp bf16x3p): a row is 128-byte blocks of 32 elements, [hi x 32 | lo x 32]; 16-byte piece sp of a row:
            // block = sp >> 3, plane = (sp >> 2) & 1, position inside the plane's 64 bytes = sp & 3  ->  staged slot block * 4 + (sp & 3)
            const char* krow = reinterpret_cast<const char*>(a.Kp) + (sh * a.nk_pad + row) * 256;
            const char* vrow = reinterpret_cast<const char*>(a.Vt) + (sh * 64 + row) * (size_t)a.nk_pad * 4;
            u32x4 kv[4], vv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                  // loads first, then the LDS stores
                const int sp = part * 4 + q;
                kv[q] = row < a.nk_pad ? *reinterpret_cast<const u32x4*>(krow + sp * 16) : zero;
                vv[q] = (sp >> 3) * 32 < a.nk_pad ? *reinterpret_cast<const u32x4*>(vrow + sp * 16) : zero;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sp = part * 4 + q;
                const int plane = (sp >> 2) & 1, slot = (sp >> 3) * 4 + (sp & 3);
                *reinterpret_cast<u32x4*>(Ks + plane * QA_PLANE + row * 128 + ((slot ^ qa_ksw(row)) << 4)) = kv[q];
                *reinterpret_cast<u32x4*>(Vts + plane * QA_PLANE + row * 128 + ((slot ^ (row & 7)) << 4)) = vv[q];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    constexpr int NF = NK64 ? 4 : 2;                              // 16-key blocks that can hold real keys
    Frag<T> fq[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) qa_get(fq[c], Qs, rq * 128 + (((c * 4 + g) ^ (rq & 7)) << 4));
    f32x4 st[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        st[f] = f32x4{0, 0, 0, 0};
        const int krow = qa_kperm(f, lr);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Frag<T> fk;
            qa_get(fk, Ks, krow * 128 + (((c * 4 + g) ^ qa_ksw(krow)) << 4));
            st[f] = mma(fk, fq[c], st[f]);
        }
    }
    const unsigned char* km = a.kmask ? a.kmask + (size_t)s * a.n_kv : nullptr;
    float pr[4 * NF];
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = qa_kperm(f, g * 4 + r);
            float sv = st[f][r];
            if (kk >= a.nk) sv = -INFINITY;                                           // tile padding: weight 0 exactly
            else if (km && kk >= a.nnull && !km[kk - a.nnull]) sv = NEG_MAX;          // masked_fill(~mask, -finfo.max), attention.py:164-168
            pr[f * 4 + r] = sv;
            mx = fmaxf(mx, sv);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ls = 0.f;
#pragma unroll
    for (int e = 0; e < 4 * NF; ++e) { pr[e] = __expf(pr[e] - mx); ls += pr[e]; }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    f32x4 o[4];
#pragma unroll
    for (int df = 0; df < 4; ++df) o[df] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int kc = 0; kc < NF / 2; ++kc) {
        Frag<T> fp;
        {
            const float p8[8] = {pr[kc * 8 + 0], pr[kc * 8 + 1], pr[kc * 8 + 2], pr[kc * 8 + 3], pr[kc * 8 + 4], pr[kc * 8 + 5], pr[kc * 8 + 6], pr[kc * 8 + 7]};
            frag_from_f32(fp, p8);
        }
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const int d = df * 16 + lr;
            Frag<T> fv;
            qa_get(fv, Vts, d * 128 + (((kc * 4 + g) ^ (d & 7)) << 4));
            o[df] = mma(fv, fp, o[df]);
        }
    }
    const float inv = 1.0f / ls;
    T* orow = reinterpret_cast<T*>(a.O) + (size_t)(m0 + rq) * a.ldo + hh * 64;
#pragma unroll
    for (int df = 0; df < 4; ++df) store4(orow + df * 16 + g * 4, o[df] * inv);
}

}  // namespace pk
using namespace pk;

// dtype 1: bf16 -- xq [S*n][ld] = LayerNorm(x), xkv [S*n][ld] = x (both bf16); wq [h*64][ldw], wkv [2*h*64][ldw] bf16 with K zero-padded to a
// multiple of 64; O bf16.  dtype 2: split-bf16 -- xq / xkv f32 rows, weights = host-split planes (4-byte units, K padded to 32), O f32.
// bias [h][n][n] f32 (or NULL), slopes [h] with causal; O [S*n][ldo] receives softmax(q k^T + bias) v with the heads merged (column hh*64 + d).
// n <= 64, no null keys, no key mask.  q_ln_s != NULL: LayerNorm folded into to_q (xq = x).
extern "C" int pk_qkv_attn(int dtype, const void* xq, const void* xkv, int ld, const void* wq, const void* wkv, int ldw, int S, int n, int h,
                           int K, const float* q_scale, const float* k_scale, float scale, const float* bias, long bias_hstride,
                           int bias_ld, const float* slopes, int causal, void* O, int ldo, const float* q_ln_s, void* stream) {
    if (dtype != 1 && dtype != 2) return PK_EINVAL;
    if (!xq || !xkv || !wq || !wkv || !q_scale || !k_scale || !O || S <= 0 || n <= 0 || n > 64 || h <= 0 || K <= 0) return PK_EINVAL;
    auto mis = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) != 0; };
    const int q = dtype == 1 ? 7 : 3, bk = dtype == 1 ? 64 : 32, esz = dtype == 1 ? 2 : 4;
    if ((K & q) || (ld & q) || (ldw & q) || (ldo & 3) || mis(xq) || mis(xkv) || mis(wq) || mis(wkv) || mis(q_scale) || mis(k_scale) ||
        (reinterpret_cast<uintptr_t>(O) & (dtype == 1 ? 7 : 15)) || (bias && mis(bias))) return PK_EALIGN;
    if (ldw < (K + bk - 1) / bk * bk) return PK_EINVAL;         // W zero-padded along K to the k-tile
    const long M = (long)S * n;
    if ((size_t)M * ld * esz >= 0xFFFFFFF0ull || (size_t)2 * h * 64 * ldw * esz >= 0xFFFFFFF0ull) return PK_EINVAL;
    QkvAttnArgs a;
    a.xq = xq; a.xkv = xkv; a.wq = wq; a.wkv = wkv; a.ld = ld; a.ldw = ldw;
    a.S = S; a.n = n; a.h = h; a.K = K; a.spt = 64 / n;
    a.q_scale = q_scale; a.k_scale = k_scale; a.scale = scale;
    a.bias = bias; a.bias_hstride = bias_hstride; a.bias_ld = bias_ld;
    a.slopes = slopes; a.causal = causal;
    a.O = O; a.ldo = ldo;
    a.recip = (65536u + (uint32_t)n - 1u) / (uint32_t)n;
    a.q_ln_s = q_ln_s;
    if (q_ln_s && mis(q_ln_s)) return PK_EALIGN;
    const int tiles = (S + a.spt - 1) / a.spt;
    dim3 grid(8 * ((tiles + 7) / 8) * h);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 1) hipLaunchKernelGGL(qkv_attn_kernel<bf16>, grid, dim3(256), QA_SMEM, st, a);
    else hipLaunchKernelGGL(qkv_attn_kernel<bf16x3>, grid, dim3(256), QA_SMEM, st, a);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// Cross-attention of S sequences of n tokens (n % 64 == 0) against cached K^ / V^T images of nk = nnull + n_kv <= 64 keys
// (layouts of pk_attn_prep with nk_pad = pk_attn_pads(...); dtype 2: its pre-split bf16x3p images): O [S*n][ldo] (bf16 / f32) <-
// softmax(l2norm(xq Wq^T) K^^T) V, heads merged.  kmask [S][n_kv] uint8 over the real keys or NULL; q_ln_s: LayerNorm folded into to_q
// (xq = the un-normalised rows) or NULL.  Operand types per dtype as for pk_qkv_attn.
extern "C" int pk_q_attn_cached(int dtype, const void* xq, int ld, const void* wq, int ldw, int S, int n, int h, int K, const float* q_scale,
                                float scale, const float* q_ln_s, const void* Kp, const void* Vt, int nk_pad, int n_kv, int nnull,
                                const unsigned char* kmask, void* O, int ldo, void* stream) {
    if (dtype != 1 && dtype != 2) return PK_EINVAL;
    if (!xq || !wq || !q_scale || !Kp || !Vt || !O || S <= 0 || n <= 0 || (n & 63) || h <= 0 || K <= 0 || n_kv <= 0 || nnull < 0) return PK_EINVAL;
    const int nk = nnull + n_kv;
    if (nk > 64 || nk_pad < nk || nk_pad > 64 || (nk_pad & 31)) return PK_EINVAL;
    auto mis = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) != 0; };
    const int q = dtype == 1 ? 7 : 3, bk = dtype == 1 ? 64 : 32, esz = dtype == 1 ? 2 : 4;
    if ((K & q) || (ld & q) || (ldw & q) || (ldo & 3) || mis(xq) || mis(wq) || mis(q_scale) || mis(Kp) || mis(Vt) ||
        (reinterpret_cast<uintptr_t>(O) & (dtype == 1 ? 7 : 15)) || (q_ln_s && mis(q_ln_s))) return PK_EALIGN;
    if (dtype == 2 && ((reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vt)) & 127)) return PK_EALIGN;      // 128-byte blocks
    if (ldw < (K + bk - 1) / bk * bk) return PK_EINVAL;
    const long M = (long)S * n;
    if ((size_t)M * ld * esz >= 0xFFFFFFF0ull || (size_t)h * 64 * ldw * esz >= 0xFFFFFFF0ull) return PK_EINVAL;
    QAttnCachedArgs a{xq, wq, ld, ldw, S, n, h, K, q_scale, scale, q_ln_s, Kp, Vt, nk_pad, nk, nnull, kmask, n_kv, O, ldo};
    const int tiles = (int)(M / 64);
    dim3 grid(8 * ((tiles + 7) / 8) * h);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 1) {
        constexpr int lds = QaTile<bf16>::SMEM + 2 * QA_PLANE;
        if (nk_pad > 32) hipLaunchKernelGGL((q_attn_cached_kernel<bf16, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((q_attn_cached_kernel<bf16, false>), grid, dim3(256), lds, st, a);
    } else {
        constexpr int lds = QaTile<bf16x3>::SMEM + 4 * QA_PLANE;           // 64 KB
        if (nk_pad > 32) hipLaunchKernelGGL((q_attn_cached_kernel<bf16x3, true>), grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL((q_attn_cached_kernel<bf16x3, false>), grid, dim3(256), lds, st, a);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}
