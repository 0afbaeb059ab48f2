// pk_gemm: every nn.Linear on the hot path (reference attention.py:50-52,117-119, cvivit.py:276,283,
// 328,333, phenaki_pytorch.py:147) as one MFMA GEMM with a fused epilogue:
//   C = act(A @ W^T + bias) (+ residual)        act in {none, GEGLU pair, LeakyReLU(0.1)}
// Roofline: MFMA-bound (bf16 2.5 PFLOP/s dense, exact-f32 157 TFLOP/s); algorithmic flops 2*M*N*K.
// Two main loops share the epilogue: gemm_core.hpp (register-staged, converts f32 A on the fly) and
// gemm_dma.hpp (LDS-DMA ring, A and W of the same type, W zero-padded along K to the k-tile).
#include "gemm_dma.hpp"
#include "gemm_p8.hpp"

namespace pk {

enum { ACT_NONE = 0, ACT_GEGLU = 1, ACT_LEAKY = 2 };

struct GemmEpilogue {
    const float* bias;   // [N] or null
    const float* res;    // [M][ldr] f32 or null (added after the activation)
    void* C;             // [M][ldc]  f32 or T   (GEGLU: [M][N/2])
    int ldr, ldc;
    int out_f32;         // 1: C is f32, 0: C is T
    int act;
    int vec_ok;          // N, ldc, ldr multiples of 4 and pointers 16-B aligned -> vector epilogue
    void* C2;            // optional second copy of an f32 result in T (bf16): the next GEMM's LDS-DMA operand (vector epilogue only)
    int ldc2;
    // LayerNorm folded into this GEMM (A holds the UN-normalised rows x, W holds gamma (.) W):
    //   LN(x) W^T = rstd * (x (gamma.W)^T - mean * s) + t,   s[n] = sum_k gamma[k] W[n][k],  t[n] = sum_k beta[k] W[n][k]
    // mean / rstd come from the A fragments of the main loop (GemmDma::run_stats); null ln_s: plain GEMM
    const float* ln_s; const float* ln_t; float ln_eps;
    // scattered f32 output (vector epilogue): element (m, n) goes to C[row_off[m] + col_off[n]] instead of C[m*ldc + n]; the 4 columns a
    // lane owns must stay contiguous (col_off[n + r] = col_off[n] + r for n % 4 == 0).  The un-patchify of cvivit.py:326-334 is exactly
    // such a separable map, so to_pixels writes the (B, C, F, H, W) video directly and the 100 MB pixel matrix never exists.
    const int* row_off; const int* col_off;
    // row statistics handed from the GEMM that PRODUCES a residual-stream row to the LayerNorm-folded GEMM that consumes it, so neither an
    // ln_rows launch nor in-loop statistics are needed: every wave writes (sum, sum of squares) of its 32 columns of row m -- of the values
    // as the consumer will read them (the T-rounded C2 copy when there is one) -- to stats_out[m][chunk][2], chunk = column / 32; the
    // consumer (ln_s set, ln_stats set) adds the ceil(K / 32) partials of a row in index order: deterministic, no atomics.
    float* stats_out; const float* ln_stats; int stats_np;
    // dup_rows > 0: every output row m is ALSO written at row m + dup_rows (C and C2).  The cond | null halves of a classifier-free-
    // guidance batch are identical until the first cross-attention, so the first self-attention block runs on half the sequences and
    // its to_out GEMM writes both copies of the residual stream.
    int dup_rows;
};

constexpr int STATS_CHUNK = 32;      // columns per partial: the 16 * TN columns one wave owns in every TN = 2 kernel

// scalar epilogue (N not a multiple of 4, e.g. heads = 2 or a 1-wide critic head): no LayerNorm fold, C2, scatter or stats_out (host refuses)
template <typename T, int TM, int TN, int WN>
__device__ __forceinline__ void gemm_epilogue_scalar(const f32x4 (&acc)[TM][TN], int M, int N, const GemmEpilogue& e, int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN, g = lane >> 4, lr = lane & 15;
    float* Cf = reinterpret_cast<float*>(e.C);
    T* Ct = reinterpret_cast<T*>(e.C);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 16 * TM + i * 16 + lr;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 16 * TN + j * 16 + g * 4;
            const f32x4 v = acc[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n + r;
                if (nn >= N) break;
                float x = v[r] + (e.bias ? e.bias[nn] : 0.f);
                if (e.act == ACT_GEGLU) {
                    if (r & 1) continue;
                    const float gate = (nn + 1 < N) ? v[r + 1] + (e.bias ? e.bias[nn + 1] : 0.f) : 0.f;
                    x = gelu_for<T>(gate) * x;
                    const size_t o = (size_t)m * e.ldc + (nn >> 1);
                    if (e.out_f32) Cf[o] = x; else store_elem(Ct + o, x);
                    continue;
                }
                if (e.act == ACT_LEAKY) x = x > 0.f ? x : 0.1f * x;
                if (e.res) x += e.res[(size_t)m * e.ldr + nn];
                const size_t o = (size_t)m * e.ldc + nn;
                if (e.out_f32) Cf[o] = x; else store_elem(Ct + o, x);
            }
        }
    }
}

// LNF: LayerNorm fold with the row statistics rsum / rsq (of the main loop, or of the producer: load_row_stats).
// Every load of the epilogue (bias / s / t vectors, the residual tile, the scatter maps) is issued up front with clamped, unconditional
// addresses and consumed afterwards: written as one load -> use -> store chain per 16x16 block the compiler waits for each load in turn
// (stores may alias the next load), i.e. TM*TN dependent L2 round trips at the tail of a kernel whose tiles all finish together.
// rows blocks (of 16 rows) per load round of the epilogue: IB * TN residual vectors in flight per lane
template <int TM> constexpr int epi_rows_per_round() { return TM > 2 ? 2 : TM; }

template <typename T, int TM, int TN, int WN = 2, bool LNF = false>
__device__ __forceinline__ void gemm_epilogue(const f32x4 (&acc)[TM][TN], int M, int N, const GemmEpilogue& e, int m0, int n0,
                                              const float* rsum = nullptr, const float* rsq = nullptr, int K = 1) {
    if (!e.vec_ok) {
        if (!LNF) gemm_epilogue_scalar<T, TM, TN, WN>(acc, M, N, e, m0, n0);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN, g = lane >> 4, lr = lane & 15;
    float* __restrict__ Cf = reinterpret_cast<float*>(e.C);
    T* __restrict__ Ct = reinterpret_cast<T*>(e.C);
    const int mb = m0 + wm * 16 * TM + lr, nb = n0 + wn * 16 * TN + g * 4;      // + i*16, + j*16
    // N % 4 == 0 and N >= 4 here, so N - 4 is a valid clamped column; values loaded through a clamped index are never stored
    f32x4 b4[TN], s4[TN], t4[TN];
    int coff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nj = min(nb + j * 16, N - 4);
        b4[j] = e.bias ? *reinterpret_cast<const f32x4*>(e.bias + nj) : f32x4{0, 0, 0, 0};
        if (LNF) {
            s4[j] = *reinterpret_cast<const f32x4*>(e.ln_s + nj);
            t4[j] = *reinterpret_cast<const f32x4*>(e.ln_t + nj);
        }
        coff[j] = e.row_off ? e.col_off[nj] : nj;
    }
    constexpr int IB = epi_rows_per_round<TM>();
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += IB) {
        // ---- phase 1: loads
        f32x4 r4[IB][TN];
        int roff[IB];
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
            const int mi = min(mb + (i0 + ii) * 16, M - 1);
            roff[ii] = e.row_off ? e.row_off[mi] : 0;
            if (!LNF && e.res && e.act != ACT_GEGLU) {      // (a folded GEMM with a residual -- no such call on the hot path -- loads it in phase 2)
#pragma unroll
                for (int j = 0; j < TN; ++j) r4[ii][j] = *reinterpret_cast<const f32x4*>(e.res + (size_t)mi * e.ldr + min(nb + j * 16, N - 4));
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) r4[ii][j] = f32x4{0, 0, 0, 0};
            }
        }
        // ---- phase 2: math and stores
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
            const int i = i0 + ii, m = mb + i * 16;
            float mean = 0.f, rstd = 1.f;
            if (LNF) {                                      // statistics of row m over its K features (biased variance, eps inside the sqrt)
                const float inv_k = 1.0f / (float)K;        // (uniform: one scalar division)
                mean = rsum[i] * inv_k;
                rstd = __builtin_amdgcn_rsqf(fmaxf(rsq[i] * inv_k - mean * mean, 0.f) + e.ln_eps);      // v_rsq_f32: 1 ulp
            }
            float ps = 0.f, pq = 0.f;                       // stats_out partial of this lane's 4 * TN columns of row m
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nb + j * 16;
                const bool live = m < M && n < N;
                f32x4 v = acc[i][j];
                if (LNF) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rstd * (v[r] - mean * s4[j][r]) + t4[j][r];
                }
                v += b4[j];
                if (e.act == ACT_GEGLU) {
                    const float o0 = gelu_for<T>(v[1]) * v[0], o1 = gelu_for<T>(v[3]) * v[2];
                    const size_t o = (size_t)m * e.ldc + (n >> 1);
                    if (live) { if (e.out_f32) store2(Cf + o, o0, o1); else store2(Ct + o, o0, o1); }
                    continue;
                }
                if (e.act == ACT_LEAKY) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
                }
                if (!LNF) v += r4[ii][j];
                else if (e.res && live) v += *reinterpret_cast<const f32x4*>(e.res + (size_t)m * e.ldr + n);
                if (live) {
                    const size_t o = e.row_off ? (size_t)((long)roff[ii] + (long)coff[j]) : (size_t)m * e.ldc + n;
                    if (e.out_f32) store4(Cf + o, v); else store4(Ct + o, v);
                    if (e.C2) store4(reinterpret_cast<bf16*>(e.C2) + (size_t)m * e.ldc2 + n, v);
                    if (e.dup_rows) {
                        const size_t od = o + (size_t)e.dup_rows * e.ldc;
                        if (e.out_f32) store4(Cf + od, v); else store4(Ct + od, v);
                        if (e.C2) store4(reinterpret_cast<bf16*>(e.C2) + (size_t)(m + e.dup_rows) * e.ldc2 + n, v);
                    }
                    if (TN == 2 && e.stats_out) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x = (sizeof(T) == 2 && (e.C2 || !e.out_f32)) ? bf2f(f2bf(v[r])) : v[r];
                            ps += x; pq += x * x;
                        }
                    }
                }
            }
            if (TN == 2 && e.stats_out) {                   // uniform branch; the 4 lane groups g hold the 4 column quarters of the chunk
                ps += __shfl_xor(ps, 16); pq += __shfl_xor(pq, 16);
                ps += __shfl_xor(ps, 32); pq += __shfl_xor(pq, 32);
                const int chunk = (n0 + wn * 16 * TN) / STATS_CHUNK;
                if (g == 0 && m < M && chunk < e.stats_np)
                    reinterpret_cast<float2*>(e.stats_out)[(size_t)m * e.stats_np + chunk] = float2{ps, pq};
            }
        }
    }
}

template <typename T, typename TA, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmOperands p, const GemmEpilogue e) {
    using Tile = GemmTile<T, TA, TM, TN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m0 = blockIdx.x * Tile::BM, n0 = blockIdx.y * Tile::BN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    Tile::run(p, m0, n0, smem, acc);
    gemm_epilogue<T, TM, TN>(acc, p.M, p.N, e, m0, n0);
}

// waves per SIMD the LDS footprint allows (160 KB per CU, 4 SIMDs): the register allocator is held to that occupancy, so the batched
// loads of the epilogue cannot push the 64x64 kernel from 5 to 4 workgroups per CU
template <typename Tile>
constexpr int lds_waves_per_simd() {
    const int wgs = 163840 / Tile::SMEM, w = wgs * (Tile::THREADS / 64) / 4;
    return w < 1 ? 1 : (w > 8 ? 8 : w);
}

template <typename T, int TM, int TN, int WM, int WN, int STAGES, int ROWB, int PW, int LNF = 0>
__global__ __launch_bounds__(64 * (WM * WN + PW))
__attribute__((amdgpu_waves_per_eu(lds_waves_per_simd<GemmDma<T, TM, TN, WM, WN, STAGES, ROWB, PW>>(), 8)))
void gemm_dma_kernel(const GemmOperands p, const GemmEpilogue e, int a_nrows) {
    using Tile = GemmDma<T, TM, TN, WM, WN, STAGES, ROWB, PW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware tile map.  Workgroup b is observed to run on XCD b % 8 (speed only, never correctness); each XCD has a
    // private 4 MiB L2.  XCD x owns a contiguous chunk of m-tiles and walks ALL n-tiles for it (m fastest), so its
    // slice of A stays L2-resident while W streams through once per XCD, instead of every XCD thrashing on all of A.
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, cmax = (MT + 7) / 8, NTn = (p.N + Tile::BN - 1) / Tile::BN;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mstart = xcd * MT / 8, mcount = (xcd + 1) * MT / 8 - mstart;
    int m0, n0;
    if (p.plain_map) {                                    // A/B reference order: m fastest over the whole grid
        if ((int)blockIdx.x >= MT * NTn) return;
        m0 = (blockIdx.x % MT) * Tile::BM;
        n0 = (blockIdx.x / MT) * Tile::BN;
    } else {
        int ml, ntile;
        if (!xcd_panel_tile(idx, cmax, mcount, NTn, p.panel, ml, ntile)) return;      // whole workgroup exits before any barrier
        m0 = (mstart + ml) * Tile::BM;
        n0 = ntile * Tile::BN;
    }
    PK_TL_KERNEL(0);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    if constexpr (LNF == 1) {
        float rsum[TM], rsq[TM];
        if (!Tile::template run_stats<2>(p, a_nrows, m0, n0, smem, acc, rsum, rsq)) return;
        gemm_epilogue<T, TM, TN, WN, true>(acc, p.M, p.N, e, m0, n0, rsum, rsq, p.K);
    } else if constexpr (LNF == 2) {
        // statistics of the tile's BM rows from the partials their producer left in e.ln_stats: 4 threads per row, each with a quarter of
        // the row's partials.  The loads are issued here and ride under the main loop (whose first vmcnt(0) covers them); sums, the
        // 4-lane fold and the hand-over through LDS happen after it.  (Waiting for them up front cost 12 us per launch: ~1.5 tiles per
        // CU slot, each starting with dependent L2 round trips; 16 dependent loads per lane in the epilogue cost 45 us.)
        constexpr int TPR = Tile::THREADS / Tile::BM;      // threads per tile row: 4 (64x64 / 8-wave 128x128) or 2 (4-wave 128x128, split-bf16)
        static_assert(Tile::THREADS == TPR * Tile::BM && (TPR == 2 || TPR == 4), "2 or 4 threads per tile row");
        constexpr int NPRE = 16 / TPR;                     // partials prefetched per thread: all of them for K = 512 (16 chunks of 32)
        const int srow = threadIdx.x / TPR, sq4 = threadIdx.x % TPR;
        const int per = (e.stats_np + TPR - 1) / TPR, c0 = sq4 * per, c1 = min(c0 + per, e.stats_np);
        const float2* st = reinterpret_cast<const float2*>(e.ln_stats) + (size_t)min(m0 + srow, p.M - 1) * e.stats_np;
        float su = 0.f, sq = 0.f;
        for (int c = c0 + NPRE; c < c1; ++c) { const float2 pr = st[c]; su += pr.x; sq += pr.y; }      // K > 512 only
        float2 pre[NPRE];
#pragma unroll
        for (int u = 0; u < NPRE; ++u) pre[u] = c0 + u < c1 ? st[c0 + u] : float2{0.f, 0.f};
        if (!Tile::run(p, a_nrows, m0, n0, smem, acc)) return;
#pragma unroll
        for (int u = 0; u < NPRE; ++u) { su += pre[u].x; sq += pre[u].y; }
        su += __shfl_xor(su, 1); sq += __shfl_xor(sq, 1);
        if (TPR == 4) { su += __shfl_xor(su, 2); sq += __shfl_xor(sq, 2); }
        __syncthreads();                                  // the last k-tile's fragments have been read: the ring can be reused
        float2* ls = reinterpret_cast<float2*>(smem);
        if (sq4 == 0) ls[srow] = float2{su, sq};
        __syncthreads();
        float rsum[TM], rsq[TM];
        const int lane = threadIdx.x & 63, wm = (threadIdx.x >> 6) / WN;
#pragma unroll
        for (int i = 0; i < TM; ++i) { const float2 pr = ls[wm * 16 * TM + i * 16 + (lane & 15)]; rsum[i] = pr.x; rsq[i] = pr.y; }
        gemm_epilogue<T, TM, TN, WN, true>(acc, p.M, p.N, e, m0, n0, rsum, rsq, p.K);
    } else {
        if (!Tile::run(p, a_nrows, m0, n0, smem, acc)) return;      // producer waves hold no accumulators
        PK_TL_KERNEL(1);
        gemm_epilogue<T, TM, TN, WN>(acc, p.M, p.N, e, m0, n0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PK_TL_KERNEL(2);
    }
}

// split-K form of the plain LDS-DMA GEMM (training: dW = dY^T X contracts over the ROWS of the batch -- K = 4608 .. 36 864 against 64 .. 344
// output tiles, so a single launch leaves most CUs idle): grid.y slices the contraction, slice z reads A / W from column z * Kc on and writes
// its partial product to C + z * M * ldc; the caller adds the slices in index order (pk_sum_batch: deterministic).  A separate kernel so the
// inference GEMMs keep their register budget.
template <typename T, int TM, int TN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(64 * WM * WN)
__attribute__((amdgpu_waves_per_eu(lds_waves_per_simd<GemmDma<T, TM, TN, WM, WN, STAGES, 128, 0>>(), 8)))
void gemm_dma_splitk_kernel(GemmOperands p, GemmEpilogue e, int a_nrows, long batch_a, long batch_w, long batch_c) {
    using Tile = GemmDma<T, TM, TN, WM, WN, STAGES, 128, 0>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int z = blockIdx.y;
    p.A = reinterpret_cast<const char*>(p.A) + (size_t)z * batch_a;
    p.W = reinterpret_cast<const char*>(p.W) + (size_t)z * batch_w;
    e.C = reinterpret_cast<char*>(e.C) + (size_t)z * batch_c;
    if (z != 0) e.bias = nullptr;                         // a bias rides on slice 0 only: the slices' sum then carries it once
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, NTn = (p.N + Tile::BN - 1) / Tile::BN;
    if ((int)blockIdx.x >= MT * NTn) return;
    const int m0 = (blockIdx.x % MT) * Tile::BM, n0 = (blockIdx.x / MT) * Tile::BN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    if (!Tile::run(p, a_nrows, m0, n0, smem, acc)) return;
    gemm_epilogue<T, TM, TN, WN>(acc, p.M, p.N, e, m0, n0);
}

template <typename T, int TM = 2, int TN = 2>
static int launch_splitk(const GemmOperands& p, const GemmEpilogue& e, int a_nrows, int splits, long ba, long bw, long bc, hipStream_t s) {
    using Tile = GemmDma<T, TM, TN, 2, 2, 2, 128, 0>;        // 4 waves as 2 x 2: 64 x 64 (TM = TN = 2) or 128 x 128 (TM = TN = 4) tiles
    constexpr int lds = Tile::SMEM;
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, NT = (p.N + Tile::BN - 1) / Tile::BN;
    hipLaunchKernelGGL((gemm_dma_splitk_kernel<T, TM, TN, 2, 2, 2>), dim3(MT * NT, splits), dim3(Tile::THREADS), lds, s, p, e, a_nrows, ba, bw, bc);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

template <typename T, typename TA, int TM, int TN>
static int launch_v1(const GemmOperands& p, const GemmEpilogue& e, hipStream_t s) {
    using Tile = GemmTile<T, TA, TM, TN>;
    dim3 grid((p.M + Tile::BM - 1) / Tile::BM, (p.N + Tile::BN - 1) / Tile::BN);
    hipLaunchKernelGGL((gemm_kernel<T, TA, TM, TN>), grid, dim3(256), Tile::SMEM, s, p, e);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

template <typename T, int TM, int TN, int STAGES, int WM = 2, int WN = 2, int ROWB = 128, int PW = 0, int LNF = 0>
static int launch_dma(const GemmOperands& p, const GemmEpilogue& e, int a_nrows, hipStream_t s) {
    using Tile = GemmDma<T, TM, TN, WM, WN, STAGES, ROWB, PW>;
    if (Tile::SMEM > 65536) {                              // opt-in to > 64 KB of LDS: per kernel AND per device of the process
        static bool attr_set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PK_ELAUNCH;
        if (!attr_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dma_kernel<T, TM, TN, WM, WN, STAGES, ROWB, PW, LNF>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, Tile::SMEM) != hipSuccess) return PK_ELAUNCH;
            attr_set[dev] = true;
        }
    }
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, NT = (p.N + Tile::BN - 1) / Tile::BN;
    dim3 grid(8 * ((MT + 7) / 8) * NT);                   // see the XCD-aware tile map in the kernel
    GemmOperands pp = p;
    static const int panel_env = [] { const char* e_ = getenv("PK_GEMM_PANEL"); return e_ ? atoi(e_) : -1; }();      // tuning knob: 0 = off, n = m-tiles per panel
    pp.panel = panel_env >= 0 ? panel_env : xcd_panel_rows(Tile::BM, p.K, (int)sizeof(T));
    hipLaunchKernelGGL((gemm_dma_kernel<T, TM, TN, WM, WN, STAGES, ROWB, PW, LNF>), grid, dim3(Tile::THREADS), Tile::SMEM, s, pp, e, a_nrows);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// 256 x 256 8-phase main loop (gemm_p8.hpp), one workgroup per CU; the four quadrants of a wave go through the ordinary epilogue
template <typename T, int FLAGS, int PH> struct P8Pick { typedef GemmP8<T, FLAGS> type; };
template <typename T, int FLAGS> struct P8Pick<T, FLAGS, 4> { typedef GemmP4<T, FLAGS> type; };

template <typename T, int FLAGS = 0, int PH = 8, int LNF = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_p8_kernel(const GemmOperands p, const GemmEpilogue e, int a_nrows) {
    using Tile = typename P8Pick<T, FLAGS, PH>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, cmax = (MT + 7) / 8, NTn = (p.N + Tile::BN - 1) / Tile::BN;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mstart = xcd * MT / 8, mcount = (xcd + 1) * MT / 8 - mstart;
    int m0, n0;
    if (p.plain_map) {
        if ((int)blockIdx.x >= MT * NTn) return;
        m0 = (blockIdx.x % MT) * Tile::BM;
        n0 = (blockIdx.x / MT) * Tile::BN;
    } else {
        int ml, ntile;
        if (!xcd_panel_tile(idx, cmax, mcount, NTn, p.panel, ml, ntile)) return;
        m0 = (mstart + ml) * Tile::BM;
        n0 = ntile * Tile::BN;
    }
    typename Tile::Acc acc;
    if constexpr ((FLAGS & 32) != 0) {                    // timing experiment (32x32x16 MFMA on the same operand registers): no real output
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][i][r] = 0.f;
        Tile::run(p, a_nrows, m0, n0, smem, acc);
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t += acc[a][b][i][r];
        if (m0 + (int)threadIdx.x < p.M) reinterpret_cast<float*>(e.C)[(size_t)(m0 + threadIdx.x) * e.ldc + n0] = t;
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0, 0, 0, 0};
        if constexpr (LNF == 2) {
            // LayerNorm folded into this GEMM with the row statistics its producer left (e.ln_stats; see gemm_dma_kernel): 512 threads = 2 per tile row,
            // each prefetches half of the row's partials ahead of the main loop; sums, the pair fold and the hand-over through LDS happen after it
            constexpr int NPRE = 8;
            const int srow = threadIdx.x >> 1, sq2 = threadIdx.x & 1;
            const int per = (e.stats_np + 1) / 2, c0 = sq2 * per, c1 = min(c0 + per, e.stats_np);
            const float2* st = reinterpret_cast<const float2*>(e.ln_stats) + (size_t)min(m0 + srow, p.M - 1) * e.stats_np;
            float su = 0.f, sq = 0.f;
            for (int c = c0 + NPRE; c < c1; ++c) { const float2 pr = st[c]; su += pr.x; sq += pr.y; }      // K > 512 only
            float2 pre[NPRE];
#pragma unroll
            for (int u = 0; u < NPRE; ++u) pre[u] = c0 + u < c1 ? st[c0 + u] : float2{0.f, 0.f};
            Tile::run(p, a_nrows, m0, n0, smem, acc);         // ends with a workgroup barrier: the ring is dead
#pragma unroll
            for (int u = 0; u < NPRE; ++u) { su += pre[u].x; sq += pre[u].y; }
            su += __shfl_xor(su, 1); sq += __shfl_xor(sq, 1);
            float2* ls = reinterpret_cast<float2*>(smem);
            if (sq2 == 0) ls[srow] = float2{su, sq};
            __syncthreads();
            const int lane = threadIdx.x & 63, wr = threadIdx.x >> 8;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float rsum[4], rsq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float2 pr = ls[a * 128 + wr * 64 + i * 16 + (lane & 15)]; rsum[i] = pr.x; rsq[i] = pr.y; }
                gemm_epilogue<T, 4, 2, 4, true>(acc[a][0], p.M, p.N, e, m0 + a * 128, n0, rsum, rsq, p.K);
                gemm_epilogue<T, 4, 2, 4, true>(acc[a][1], p.M, p.N, e, m0 + a * 128, n0 + 128, rsum, rsq, p.K);
            }
        } else {
            Tile::run(p, a_nrows, m0, n0, smem, acc);
            // (four explicit calls: inside a loop over (a, b) hipcc keeps the accumulators in scratch for the epilogue)
            gemm_epilogue<T, 4, 2, 4>(acc[0][0], p.M, p.N, e, m0, n0);
            gemm_epilogue<T, 4, 2, 4>(acc[0][1], p.M, p.N, e, m0, n0 + 128);
            gemm_epilogue<T, 4, 2, 4>(acc[1][0], p.M, p.N, e, m0 + 128, n0);
            gemm_epilogue<T, 4, 2, 4>(acc[1][1], p.M, p.N, e, m0 + 128, n0 + 128);
        }
    }
}

template <typename T, int FLAGS = 0, int PH = 8, int LNF = 0>
static int launch_p8(const GemmOperands& p, const GemmEpilogue& e, int a_nrows, hipStream_t s) {
    using Tile = typename P8Pick<T, FLAGS, PH>::type;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PK_ELAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_kernel<T, FLAGS, PH, LNF>), hipFuncAttributeMaxDynamicSharedMemorySize, Tile::SMEM) != hipSuccess) return PK_ELAUNCH;
        attr_set[dev] = true;
    }
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, NT = (p.N + Tile::BN - 1) / Tile::BN;
    dim3 grid(8 * ((MT + 7) / 8) * NT);
    GemmOperands pp = p;
    static const int panel_env = [] { const char* e_ = getenv("PK_GEMM_PANEL"); return e_ ? atoi(e_) : -1; }();
    pp.panel = panel_env >= 0 ? panel_env : xcd_panel_rows(Tile::BM, p.K, (int)sizeof(T));
    hipLaunchKernelGGL((gemm_p8_kernel<T, FLAGS, PH, LNF>), grid, dim3(Tile::THREADS), Tile::SMEM, s, pp, e, a_nrows);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// split-K form of the two-group 256 x 256 loop (bf16; pk_gemm_splitk tile = 2): grid.y slices the contraction exactly like gemm_dma_splitk_kernel; for products whose
// 256 x 256 tiles are far fewer than the CUs while K is long (the patch-embedding shape 4096 x 512 x 6144: 32 tiles x 8 slices of 12 k-tiles)
template <typename T>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_p8_splitk_kernel(GemmOperands p, GemmEpilogue e, int a_nrows, long batch_a, long batch_w, long batch_c) {
    using Tile = GemmP4<T, 0>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int z = blockIdx.y;
    p.A = reinterpret_cast<const char*>(p.A) + (size_t)z * batch_a;
    p.W = reinterpret_cast<const char*>(p.W) + (size_t)z * batch_w;
    e.C = reinterpret_cast<char*>(e.C) + (size_t)z * batch_c;
    if (z != 0) e.bias = nullptr;
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, NTn = (p.N + Tile::BN - 1) / Tile::BN;
    if ((int)blockIdx.x >= MT * NTn) return;
    const int m0 = (blockIdx.x % MT) * Tile::BM, n0 = (blockIdx.x / MT) * Tile::BN;
    typename Tile::Acc acc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0, 0, 0, 0};
    Tile::run(p, a_nrows, m0, n0, smem, acc);
    gemm_epilogue<T, 4, 2, 4>(acc[0][0], p.M, p.N, e, m0, n0);
    gemm_epilogue<T, 4, 2, 4>(acc[0][1], p.M, p.N, e, m0, n0 + 128);
    gemm_epilogue<T, 4, 2, 4>(acc[1][0], p.M, p.N, e, m0 + 128, n0);
    gemm_epilogue<T, 4, 2, 4>(acc[1][1], p.M, p.N, e, m0 + 128, n0 + 128);
}

static int launch_p8_splitk(const GemmOperands& p, const GemmEpilogue& e, int a_nrows, int splits, long ba, long bw, long bc, hipStream_t s) {
    using Tile = GemmP4<bf16, 0>;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PK_ELAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p8_splitk_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, Tile::SMEM) != hipSuccess) return PK_ELAUNCH;
        attr_set[dev] = true;
    }
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, NT = (p.N + Tile::BN - 1) / Tile::BN;
    hipLaunchKernelGGL((gemm_p8_splitk_kernel<bf16>), dim3(MT * NT, splits), dim3(Tile::THREADS), Tile::SMEM, s, p, e, a_nrows, ba, bw, bc);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

}  // namespace pk

using namespace pk;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the DMA ring needs A in T, W zero-padded along K to the k-tile (the packers guarantee ldw >= round_up(K, BK)),
// and both matrices below 4 GiB (32-bit buffer offsets)
static bool dma_possible(int dtype, int a_is_f32, int N, int K, int lda, int ldw, int a_nrows) {
    const int bk = dtype == 1 ? 64 : 32;
    const int kpad = (K + bk - 1) / bk * bk;
    const size_t el = dtype == 1 ? 2 : 4;
    return !(dtype == 1 && a_is_f32) && ldw >= kpad &&
           (size_t)a_nrows * lda * el < 0xFFFFFFF0ull && (size_t)N * ldw * el < 0xFFFFFFF0ull;
}

// measured on MI355X (tools/gemm_bench.py, profiles/gemm_variants_r01.txt): the loop is bound by the L2 -> LDS fill
// rate, which grows with the number of co-resident workgroups -> shallow rings; 128x128 tiles (2x the flop/byte) only
// pay once there are enough of them to fill 256 CUs twice, or when K is long
static int auto_variant(int dtype, int a_is_f32, int M, int N, int K, int lda, int ldw, int a_nrows) {
    const long blocks128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (!dma_possible(dtype, a_is_f32, N, K, lda, ldw, a_nrows)) return blocks128 >= 384 ? 2 : 1;
    // 128x128 once there are >= 256 of them (one per CU): measured in the sampling loop (M = 9216: to_out 288 tiles) +2.5 %
    // tokens/s over a 512 threshold; at 144 tiles (M = 4608, N = 512) the 64x64 kernel wins by 9 % (same-box A/B)
    static const long t128 = getenv("PK_GEMM_T128") ? atol(getenv("PK_GEMM_T128")) : 256;      // tuning knob
    // 256..383 tiles of 128x128 with a short K (the sampling loop's to_out, M = 9216, N = K = 512: 288 tiles = 1.1 per CU): 128x64
    // tiles give every CU 2+ workgroups -- 14.9 vs 17.0 us with the residual epilogue (8-wave 128x64 / 64x128 layouts: 16.5 us)
    if (dtype != 0 && blocks128 >= t128 && blocks128 < 384 && K <= 1024) return 27;
    // split-bf16: 128x128 with 4 waves (64x64 wave tiles: every in-register A split feeds 4 column fragments): 98 vs 107 us at
    // 9216 x 2736 x 512, 185 vs 208 us at 18432 rows (tools/gemm_bench.py --mode bf16x3)
    // long K and >= 2 rounds of 256 x 256 tiles (no hot-path shape; the library entry point is general): the two-group 256 x 256 loop of gemm_p8.hpp,
    // 1.32-1.37 vs 1.02-1.05 PF at 8192^3 (profiles/gemm_p8_r06.txt); at K = 512 the 128 x 128 loop's two co-resident workgroups win
    if (dtype == 1 && K >= 2048 && (long)((M + 255) / 256) * ((N + 255) / 256) >= 512) return 50;
    if (blocks128 >= t128) return dtype == 2 ? 9 : 24;                      // 128x128, 8 waves, 2 stages (16 waves/CU)
    if (K >= 2048) return dtype == 1 ? 33 : 3;      // long K (patch embed): 64x64, 3-stage ring fed by 2 producer waves (bf16) / 4 stages
    return 8;                                                               // 64x64, 2 stages (5 WG/CU)
}

// which main loop pk_gemm picks for a shape (1/2 register-staged 64x64 / 128x128; 3, 8 DMA 64x64 4 / 2 stages; 9 DMA 128x128)
extern "C" int pk_gemm_auto_variant(int dtype, int a_is_f32, int M, int N, int K, int lda, int ldw, int a_nrows) {
    return auto_variant(dtype, a_is_f32, M, N, K, lda, ldw, a_nrows);
}

// variant: 0 = automatic; 1/2 = register-staged 64x64 / 128x128; 8 = DMA 64x64 2 stages; 9 / 24 = DMA 128x128 (4 / 8 waves);
//          27 = DMA 128x64 4 waves (bf16); 33 = DMA 64x64 with 2 producer waves, 3 stages (bf16); 3 = DMA 64x64 4 stages (f32)   (explicit: tools/gemm_bench.py)
extern "C" int pk_gemm_ex(int dtype, int a_is_f32, const void* A, int lda, const void* W, int ldw,
                          int M, int N, int K, const float* bias, const float* res, int ldr,
                          void* C, int ldc, int out_is_f32, int act, const int* a_rows, int a_nrows,
                          int variant, void* C2, int ldc2, const float* ln_s, const float* ln_t, float ln_eps,
                          const int* row_off, const int* col_off, float* stats_out, const float* ln_stats, int dup_rows, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !C) return PK_EINVAL;
    if (dtype != 0 && dtype != 1 && dtype != 2) return PK_EINVAL;
    if (act < 0 || act > 2) return PK_EINVAL;
    const int eps_w = dtype == 1 ? 8 : 4;                        // elements per 16 B of W (split-bf16: W is counted in 4-byte units, like f32)
    const int eps_a = (dtype == 1 && !a_is_f32) ? 8 : 4;         // alignment quantum of A rows
    if (K % eps_w || ldw % eps_w || lda % eps_a) return PK_EALIGN;
    if (dtype == 1 && a_is_f32 && (K % 8)) return PK_EALIGN;
    if (!al16(A) || !al16(W)) return PK_EALIGN;
    if (dtype != 1 && !a_is_f32) return PK_EINVAL;               // exact-f32 and split-bf16 modes keep every activation f32
    if (dtype == 2 && C2) return PK_EINVAL;      // the bf16 copy is a bf16-mode feature (split-bf16 consumers read the f32 rows themselves)
    if (act == ACT_GEGLU && (N & 1)) return PK_EINVAL;
    if (a_rows && a_nrows <= 0) return PK_EINVAL;
    if (!a_rows) a_nrows = M;
    // k-rotation (gemm_dma.hpp): measured +12..33 % on the 65536-wide vocab-head shape, -0..13 % on the N <= 2736 shapes
    GemmOperands p{A, W, a_rows, lda, ldw, M, N, K, 0, N >= 8192 ? krot_default() : 0};
    GemmEpilogue e{bias, res, C, ldr, ldc, out_is_f32, act, 0, C2, ldc2, ln_s, ln_t, ln_eps, row_off, col_off, stats_out, ln_stats,
                   ((ln_stats ? K : N) + STATS_CHUNK - 1) / STATS_CHUNK, dup_rows};
    bool v = (N % 4 == 0) && (ldc % 4 == 0) && al16(C) && (!bias || al16(bias)) && (!res || (al16(res) && ldr % 4 == 0));
    if (act == ACT_GEGLU) v = v && (ldc % 2 == 0) && ((reinterpret_cast<uintptr_t>(C) & 7) == 0);
    e.vec_ok = v ? 1 : 0;
    if (C2 && (!v || !out_is_f32 || act == ACT_GEGLU || dtype != 1 || (ldc2 & 3) || (reinterpret_cast<uintptr_t>(C2) & 7))) return PK_EINVAL;
    if ((ln_s == nullptr) != (ln_t == nullptr)) return PK_EINVAL;
    if ((row_off == nullptr) != (col_off == nullptr)) return PK_EINVAL;
    if (row_off && (!v || !out_is_f32 || act == ACT_GEGLU || res || C2)) return PK_EINVAL;
    if (ln_stats && (!ln_s || stats_out || (reinterpret_cast<uintptr_t>(ln_stats) & 7))) return PK_EINVAL;
    if (stats_out && (!v || act == ACT_GEGLU || row_off || (reinterpret_cast<uintptr_t>(stats_out) & 7))) return PK_EINVAL;
    if (dup_rows < 0 || (dup_rows && (!v || act == ACT_GEGLU || row_off || stats_out || dup_rows < M))) return PK_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);

    const bool dma_ok = dma_possible(dtype, a_is_f32, N, K, lda, ldw, a_nrows);
    if (variant >= 100) { p.plain_map = 1; variant -= 100; }
    if (variant == 0) variant = auto_variant(dtype, a_is_f32, M, N, K, lda, ldw, a_nrows);
    if (variant >= 3 && !dma_ok) return PK_EINVAL;
    if (ln_s) {
        // LayerNorm-folded GEMM: LDS-DMA main loop only (the statistics come from its A fragments), vector epilogue, 64x64 / 128x128 tiles
        if (!dma_ok || !v || !al16(ln_s) || !al16(ln_t) || a_rows) return PK_EINVAL;
        const bool big = variant == 24 || variant == 2 || variant == 9;
#ifdef PK_P8_ABLATE
        if (ln_stats && variant == 50 && dtype != 0) {         // the two-group 256 x 256 loop with the producer's row statistics (measured, not used: see below)
            return dtype == 1 ? launch_p8<bf16, 0, 4, 2>(p, e, a_nrows, s) : launch_p8<bf16x3, 0, 4, 2>(p, e, a_nrows, s);
        }
#endif
        if (ln_stats) {
            if (dtype == 1) return big ? launch_dma<bf16, 4, 2, 2, 2, 4, 128, 0, 2>(p, e, a_nrows, s) : launch_dma<bf16, 2, 2, 2, 2, 2, 128, 0, 2>(p, e, a_nrows, s);
            // split-bf16 (round 4): the 4-wave 128x128 tile of that mode (2 threads per row fetch the partials)
            if (dtype == 2) return big ? launch_dma<bf16x3, 4, 4, 2, 2, 2, 128, 0, 2>(p, e, a_nrows, s) : launch_dma<bf16x3, 2, 2, 2, 2, 2, 128, 0, 2>(p, e, a_nrows, s);
            return big ? launch_dma<float, 4, 2, 2, 2, 4, 128, 0, 2>(p, e, a_nrows, s) : launch_dma<float, 2, 2, 2, 2, 2, 128, 0, 2>(p, e, a_nrows, s);
        }
        if (dtype == 1) return big ? launch_dma<bf16, 4, 2, 2, 2, 4, 128, 0, 1>(p, e, a_nrows, s) : launch_dma<bf16, 2, 2, 2, 2, 2, 128, 0, 1>(p, e, a_nrows, s);
        if (dtype == 2) return big ? launch_dma<bf16x3, 4, 4, 2, 2, 2, 128, 0, 1>(p, e, a_nrows, s) : launch_dma<bf16x3, 2, 2, 2, 2, 2, 128, 0, 1>(p, e, a_nrows, s);
        return big ? launch_dma<float, 4, 2, 2, 2, 4, 128, 0, 1>(p, e, a_nrows, s) : launch_dma<float, 2, 2, 2, 2, 2, 128, 0, 1>(p, e, a_nrows, s);
    }
    // stats_out is written by the TN = 2 LDS-DMA kernels only (one 32-column chunk per wave)
    if (stats_out && !(dma_ok && (variant == 8 || variant == 24 || variant == 27 || variant == 33 || variant == 50 || (variant == 3 && dtype != 1)))) return PK_EINVAL;
    // the main-loop variants that survived round 1's sweep (profiles/gemm_variants*_r01.txt; the 35 losers -- deeper rings, k-tile 32,
    // 128x256 / 256x256 tiles, other wave layouts, other producer / consumer splits -- were deleted in round 2)
    if (dtype == 1) {
        switch (variant) {
            case 1: return a_is_f32 ? launch_v1<bf16, float, 2, 2>(p, e, s) : launch_v1<bf16, bf16, 2, 2>(p, e, s);      // register-staged 64x64
            case 2: return a_is_f32 ? launch_v1<bf16, float, 4, 4>(p, e, s) : launch_v1<bf16, bf16, 4, 4>(p, e, s);      // register-staged 128x128
            case 8: return launch_dma<bf16, 2, 2, 2>(p, e, a_nrows, s);                 // 64x64, 2 stages (5 WG/CU)
            case 9: return launch_dma<bf16, 4, 4, 2>(p, e, a_nrows, s);                 // 128x128, 4 waves, 2 stages
            case 24: return launch_dma<bf16, 4, 2, 2, 2, 4>(p, e, a_nrows, s);          // 128x128, 8 waves (2x4), 2 stages (64 KB: 16 waves/CU)
            case 33: return launch_dma<bf16, 2, 2, 3, 2, 2, 128, 2>(p, e, a_nrows, s);  // 64x64, 4 consumers + 2 producers, 3 stages (long K)
            case 27: return launch_dma<bf16, 4, 2, 2, 2, 2>(p, e, a_nrows, s);          // 128x64, 4 waves (wave tile 64x32), 2 stages (48 KB: 3 WG/CU)
            case 50: return launch_p8<bf16, 0, 4>(p, e, a_nrows, s);                    // 256x256, 8 waves in two groups a barrier apart, 2 phases per k-tile (gemm_p8.hpp), 1 WG/CU
#ifdef PK_P8_ABLATE
            // tools/gemm_bench.py ablations behind profiles/gemm_p8_r06.txt (build with PK_EXTRA_HIPCC_FLAGS=-DPK_P8_ABLATE); FLAGS are listed in gemm_p8.hpp
            case 55: return launch_dma<bf16, 8, 8, 2, 2, 2>(p, e, a_nrows, s);   // ONE wave per SIMD: 256 x 256 tile, 4 waves of 128 x 128 (256 accumulator registers), the plain 2-stage ring
            case 56: return launch_dma<bf16, 8, 4, 3, 2, 2>(p, e, a_nrows, s);   // 256 x 128, 4 waves of 128 x 64, 3 stages (144 KB)
            case 60: return launch_p8<bf16, 0, 8>(p, e, a_nrows, s);        // the 8-phase form (16 MFMAs per phase)
            case 61: return launch_p8<bf16, 1, 8>(p, e, a_nrows, s);
            case 62: return launch_p8<bf16, 2, 8>(p, e, a_nrows, s);
            case 63: return launch_p8<bf16, 3, 8>(p, e, a_nrows, s);
            case 64: return launch_p8<bf16, 4, 8>(p, e, a_nrows, s);
            case 68: return launch_p8<bf16, 8, 8>(p, e, a_nrows, s);
            case 70: return launch_p8<bf16, 8, 4>(p, e, a_nrows, s);        // 2-phase form: no s_setprio
            case 71: return launch_p8<bf16, 1, 4>(p, e, a_nrows, s);        // no in-loop DMA
            case 72: return launch_p8<bf16, 2, 4>(p, e, a_nrows, s);        // no fragment reads
            case 73: return launch_p8<bf16, 3, 4>(p, e, a_nrows, s);        // neither: MFMAs + barriers
            case 74: return launch_p8<bf16, 4, 4>(p, e, a_nrows, s);        // no stagger between the wave groups
            case 75: return launch_p8<bf16, 64, 4>(p, e, a_nrows, s);       // DMA issued, never waited for (timing only)
            case 76: return launch_p8<bf16, 128, 4>(p, e, a_nrows, s);      // fragment reads waited for after the barrier
            case 79: return launch_p8<bf16, 259, 4>(p, e, a_nrows, s);      // MFMAs only: no loads, no barriers
            case 81: return launch_p8<bf16, 512, 4>(p, e, a_nrows, s);      // DMA 4 + 4 pieces per phase (timing only)
            case 83: return launch_p8<bf16, 1024, 4>(p, e, a_nrows, s);     // a phase's reads issued before its DMA pieces
            case 84: return launch_p8<bf16, 2048, 4>(p, e, a_nrows, s);     // half of the DMA pieces issued between the MFMAs
            case 85: return launch_p8<bf16, 4096 + 64, 4>(p, e, a_nrows, s);   // + 4 ordinary VGPR buffer loads per k-tile between the MFMAs (timing only; vmcnt waits off: the extra loads change the counts)
            case 88: return launch_p8<bf16, 4096 + 8192 + 64, 4>(p, e, a_nrows, s);   // + 16 of them per k-tile (what a one-wave-per-SIMD design would issue from its MFMA stream)
            case 86: return launch_p8<bf16, 32, 4>(p, e, a_nrows, s);       // v_mfma_f32_32x32x16_bf16 on the same operand registers (timing only)
            case 87: return launch_p8<bf16, 35, 4>(p, e, a_nrows, s);       // ... without loads
#endif
            // (round 5: a 256x128 "ping-pong" loop -- one workgroup per CU, two 4-wave groups half an iteration apart, 3-stage 144 KB ring, persistent --
            //  was built, measured and removed: equal to this loop at long K, 13-40 % slower at K = 512; profiles/gemm_pingpong_r05.txt)
            // (256x256 / 256x128 / 128x256 8-wave instantiations were measured again in round 3 against the torch.mm yardstick and removed:
            //  profiles/gemm_bigtile_r03.txt -- 552 vs 653 TFLOP/s on the vocabulary-head shape, 1081 vs 1011 at 8192^3)
            default: return PK_EINVAL;
        }
    }
    if (dtype == 2) {
        // split-bf16 ("bf16x3", common.hpp): f32 A rows split in registers, host-packed (hi | lo) W planes, 3 bf16 MFMAs per fragment
        // pair; LDS-DMA main loops only (the register-staged fallbacks do not exist for it)
        if (!dma_ok) return PK_EINVAL;
        switch (variant) {
            case 3: return launch_dma<bf16x3, 2, 2, 4>(p, e, a_nrows, s);                // 64x64, 4 stages (long K)
            case 8: return launch_dma<bf16x3, 2, 2, 2>(p, e, a_nrows, s);
            case 9: return launch_dma<bf16x3, 4, 4, 2>(p, e, a_nrows, s);
            case 24: return launch_dma<bf16x3, 4, 2, 2, 2, 4>(p, e, a_nrows, s);
            case 27: return launch_dma<bf16x3, 4, 2, 2, 2, 2>(p, e, a_nrows, s);
#ifdef PK_P8_ABLATE
            // the 256 x 256 two-group loop on the (hi | lo) operand images (gemm_p8.hpp, SPLIT): bit-identical to variant 24, measured in round 6 and NOT used --
            // split-bf16 is bound by its three MFMAs per product, not by the fill: 8192^3 390 vs 352 TF-equivalent, FF1 at 9216 rows 111 vs 104 us, at 4608 rows
            // 107 vs 61 us (profiles/gemm_p8_r06.txt)
            case 50: return launch_p8<bf16x3, 0, 4>(p, e, a_nrows, s);
#endif
            default: return PK_EINVAL;
        }
    }
    switch (variant) {
        case 1: return launch_v1<float, float, 2, 2>(p, e, s);
        case 2: return launch_v1<float, float, 4, 4>(p, e, s);
        case 3: return launch_dma<float, 2, 2, 4>(p, e, a_nrows, s);                    // 64x64, 4 stages (long K)
        case 8: return launch_dma<float, 2, 2, 2>(p, e, a_nrows, s);
        case 9: return launch_dma<float, 4, 4, 2>(p, e, a_nrows, s);
        case 24: return launch_dma<float, 4, 2, 2, 2, 4>(p, e, a_nrows, s);
        default: return PK_EINVAL;
    }
}

#ifdef PK_TIMELINE
extern "C" int pk_debug_timeline(unsigned long long* out, int n) {
    if (hipDeviceSynchronize() != hipSuccess) return PK_ELAUNCH;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pk_tl), sizeof(unsigned long long) * n) == hipSuccess ? PK_OK : PK_ELAUNCH;
}
extern "C" int pk_debug_timeline_clear() {
    static unsigned long long z[8 * 5 * 40] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(pk_tl), z, sizeof(z)) == hipSuccess ? PK_OK : PK_ELAUNCH;
}
#endif

extern "C" int pk_gemm(int dtype, int a_is_f32, const void* A, int lda, const void* W, int ldw,
                       int M, int N, int K, const float* bias, const float* res, int ldr,
                       void* C, int ldc, int out_is_f32, int act, const int* a_rows, void* stream) {
    // without the physical row count of a gathered A the bounds-checked DMA path is not available: register-staged
    if (a_rows) {
        const long blocks128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        return pk_gemm_ex(dtype, a_is_f32, A, lda, W, ldw, M, N, K, bias, res, ldr, C, ldc, out_is_f32, act, a_rows,
                          0x7fffffff / (lda > 0 ? lda : 1) / 4, blocks128 >= 384 ? 2 : 1, nullptr, 0, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, 0, stream);
    }
    return pk_gemm_ex(dtype, a_is_f32, A, lda, W, ldw, M, N, K, bias, res, ldr, C, ldc, out_is_f32, act, nullptr, M, 0,
                      nullptr, 0, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

// C[z] = A[:, z Kc : (z + 1) Kc] W[:, z Kc : (z + 1) Kc]^T (+ bias on slice 0) for z < splits, Kc = K / splits (a multiple of the k-tile: 64 bf16 / 32 otherwise):
// the K-slices of one product as `splits` partial (M, N) f32 matrices, C + z * M * ldc -- sum them with pk_sum_batch.  A: T (dtype 1) or f32
// (dtype 0 / 2), W: the operand image of the dtype; both K-padded as for pk_gemm.
extern "C" int pk_gemm_splitk(int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, int splits, float* C, int ldc,
                              const float* bias, int tile, void* stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || splits < 1 || splits > 64 || (dtype != 0 && dtype != 1 && dtype != 2)) return PK_EINVAL;
    if (tile != 0 && tile != 1 && !(tile == 2 && dtype == 1)) return PK_EINVAL;      // tile 2: 256 x 256 two-group loop, bf16 only
    const int bk = dtype == 1 ? 64 : 32, esz = dtype == 1 ? 2 : 4;
    if (K % (splits * bk) || (N & 3) || (ldc & 3) || ldc < N) return PK_EINVAL;
    if (!al16(A) || !al16(W) || !al16(C) || (bias && !al16(bias)) || lda % (16 / esz) || ldw % (16 / esz)) return PK_EALIGN;
    if (!dma_possible(dtype, dtype == 1 ? 0 : 1, N, K, lda, ldw, M)) return PK_EINVAL;
    const int Kc = K / splits;
    GemmOperands p{A, W, nullptr, lda, ldw, M, N, Kc, 1, 0};
    GemmEpilogue e{bias, nullptr, C, 0, ldc, 1, ACT_NONE, 1};
    const long ba = (long)Kc * esz, bw = (long)Kc * esz, bc = (long)M * ldc * 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (tile == 2) return launch_p8_splitk(p, e, M, splits, ba, bw, bc, s);
    if (tile == 1) {                                       // 128 x 128 tiles: half the fill bytes per flop; for LONG K-slices of a big (M, N)
        if (dtype == 1) return launch_splitk<bf16, 4, 4>(p, e, M, splits, ba, bw, bc, s);
        if (dtype == 2) return launch_splitk<bf16x3, 4, 4>(p, e, M, splits, ba, bw, bc, s);
        return launch_splitk<float, 4, 4>(p, e, M, splits, ba, bw, bc, s);
    }
    if (dtype == 1) return launch_splitk<bf16>(p, e, M, splits, ba, bw, bc, s);
    if (dtype == 2) return launch_splitk<bf16x3>(p, e, M, splits, ba, bw, bc, s);
    return launch_splitk<float>(p, e, M, splits, ba, bw, bc, s);
}
