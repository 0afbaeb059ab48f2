// MaskGIT sampling-step kernels (reference phenaki_pytorch.py:478-550):
//
//  pk_cfg_mix      : e = null + (cond - null) * scale on the 512-d trunk outputs.  Classifier-free guidance is
//                    linear in the logits (phenaki_pytorch.py:155-161) and to_logits is linear, so mixing BEFORE
//                    the vocab head gives the same logits with ONE 65 536-wide GEMM instead of two.
//  pk_vocab_sample : logits = e @ W^T + b never touch HBM: the GEMM epilogue adds gumbel noise
//                    (phenaki_pytorch.py:88-93), keeps per-row (best noisy value, index, raw logit) and, when no
//                    critic is used, the online-softmax (max, sum exp) needed for 1 - softmax[pred] (:547-550);
//                    one partial per (row, 128-column tile).
//  pk_vocab_reduce : folds the partials, writes pred ids, ids = where(mask, pred, ids) (:509) and the
//                    confidence scores where(mask, 1 - p, -1e4).
//  pk_topk_mask    : mask = top-k(scores) per row (:488-489) and ids = where(mask, mask_id, ids) (:491).
// Noise: PARITY mode reads U[0,1) from memory (the tests inject the oracle's draws) and uses logf / true
// division exactly like the reference expression; FAST mode draws U from the counter hash in common.hpp (four draws per
// hash chain) and compares in the log2 domain (no eps terms: a draw of exactly 0 maps to -inf and never wins).
#include "gemm_dma.hpp"

namespace pk {

__global__ __launch_bounds__(256) void cfg_mix_kernel(const float* __restrict__ x, int ldx, int nb, int n_tot, int n_prime,
                                                      const int* __restrict__ rows, int nrows, float scale, int has_null,
                                                      void* __restrict__ out, int ldo, int out_f32, int D) {
    // output row r <- logical (b, i): rows ? rows[r] (flat b*n + i over the NON-prime positions) : r
    const int dv = D >> 2;
    const int n = n_tot - n_prime;
    const long total = (long)nrows * dv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % dv) * 4;
        const int r = (int)(idx / dv);
        const int lr = rows ? rows[r] : r;
        const int b = lr / n, i = lr % n + n_prime;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)b * n_tot + i) * ldx + c);
        if (has_null) {
            const f32x4 nv = *reinterpret_cast<const f32x4*>(x + ((size_t)(nb + b) * n_tot + i) * ldx + c);
            v = nv + (v - nv) * scale;
        }
        if (out_f32) store4(reinterpret_cast<float*>(out) + (size_t)r * ldo + c, v);
        else store4(reinterpret_cast<bf16*>(out) + (size_t)r * ldo + c, v);
    }
}

struct VocabArgs {
    const float* bias;       // [V]
    const float* U;          // PARITY: uniform noise [M][V]; FAST: unused
    uint32_t philox_stride;  // PARITY without U: the noise is torch's own uniform_ stream (common.hpp torch_uniform): 256 * grid of the (rows, V) fill
    unsigned long long philox_offset;   // ... the generator's Philox offset at that fill; the seed is (seed_lo, seed_hi)
    const int* rows;         // optional: output row r is logical row rows[r] (noise / partial indexing)
    float temp;              // max(temperature, 1e-10)
    uint32_t seed_lo, seed_hi;
    const unsigned long long* seed_dev;   // optional: added to the seed at run time (a captured hipGraph replays with fresh noise)
    int need_lse;
    int no_noise;            // 1: plain argmax of the logits (cosine-sim codebook lookup of the VectorQuantize path)
    int ntiles;
    int plain_order;         // 1: (row tile fastest, vocabulary tile) dispatch order of round 2 (A/B knob PK_VOCAB_PANEL=0)
    // partials, SoA [ntiles][M]
    float* p_val; int* p_idx; float* p_logit; float* p_max; float* p_sum;
};

// vocab-head tile: 128 rows x 128 vocabulary columns, 8 waves as 2 (rows) x 4 (cols), wave tile 64 x 32, LDS-DMA ring of 2
// (64 KB -> 2 workgroups = 16 waves per CU): the best main loop for N = 65 536 in tools/gemm_bench.py that exists for
// both operand types
// (split-bf16: 4 waves as 2 x 2 with 64 x 64 wave tiles -- each A fragment's in-register split is shared by 4 instead of 2 column
// fragments; tools/gemm_bench.py: 1138 vs 1328 us on the 4608 x 65536 x 512 head)
template <typename T> struct VocabGeom { static constexpr int TM = 4, TN = 2, WM = 2, WN = 4; };
template <> struct VocabGeom<bf16x3> { static constexpr int TM = 4, TN = 4, WM = 2, WN = 2; };
template <typename T> using VocabTile = GemmDma<T, VocabGeom<T>::TM, VocabGeom<T>::TN, VocabGeom<T>::WM, VocabGeom<T>::WN, 2>;

template <typename T, bool PARITY, bool LSE>
__global__ __launch_bounds__(64 * VocabGeom<T>::WM * VocabGeom<T>::WN) void vocab_sample_kernel(const GemmOperands p, const VocabArgs e) {
    using Tile = VocabTile<T>;
    constexpr int VTM = VocabGeom<T>::TM, VTN = VocabGeom<T>::TN, VWN = VocabGeom<T>::WN;
    static_assert(Tile::BM == 128 && Tile::BN == 128, "partials are laid out per 128-column tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- XCD-aware, L2-panelled tile order.  Workgroup b is observed to run on XCD b % 8 (speed only).  XCD x owns a contiguous
    // eighth of the vocabulary tiles (its slice of W: 8.4 MB at V = 65 536) and walks it in W-panels of 16 tiles (2 MB) x A-panels of
    // 8 row tiles (1 MB), rows fastest: both panels stay in the XCD's 4 MB L2 while their 128 tiles run.  The former (row tile, vocab
    // tile) grid in dispatch order spread every vocabulary tile's row tiles over all 8 XCDs: each L2 then cycled through the WHOLE of A
    // (4.7 MB at 4608 rows) once per vocabulary tile and fetched every W tile 8 times.
    const int MT = (p.M + Tile::BM - 1) / Tile::BM;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int cn = (e.ntiles + 7) / 8;
    const int nstart = xcd * e.ntiles / 8, ncount = (xcd + 1) * e.ntiles / 8 - nstart;
    int mtile, ntile;
    if (e.plain_order) {
        mtile = (int)(blockIdx.x % (unsigned)MT); ntile = (int)(blockIdx.x / (unsigned)MT);
        if (ntile >= e.ntiles) return;
    } else {
        constexpr int PN = 16, PM = 8;
        const int full_wp = cn / PN;
        int wp = idx / (PN * MT);
        if (wp > full_wp) wp = full_wp;
        const int basen = wp * PN, pnl = wp < full_wp ? PN : cn - basen;
        const int r = idx - wp * PN * MT;
        const int full_ap = MT / PM;
        int ap = r / (pnl * PM);
        if (ap > full_ap) ap = full_ap;
        const int basem = ap * PM, pml = ap < full_ap ? PM : MT - basem;
        const int r2 = r - ap * pnl * PM;
        mtile = basem + r2 % pml;
        ntile = basen + r2 / pml;
        if (ntile >= ncount) return;                       // whole workgroup exits before any barrier
        ntile += nstart;
    }
    const int m0 = mtile * Tile::BM, n0 = ntile * Tile::BN;
    f32x4 acc[VTM][VTN];
#pragma unroll
    for (int i = 0; i < VTM; ++i)
#pragma unroll
        for (int j = 0; j < VTN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    (void)Tile::run(p, p.M, m0, n0, smem, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / VWN, wn = wave % VWN, g = lane >> 4, lr = lane & 15;
    uint32_t seed_lo = e.seed_lo, seed_hi = e.seed_hi;
    if (!PARITY && e.seed_dev) {
        const unsigned long long sd = (((unsigned long long)e.seed_hi << 32) | e.seed_lo) + *e.seed_dev;
        seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32);
    }
    // FAST mode works in the log2 domain: argmax_v (l_v / T - ln(-ln u_v)) = argmax_v (l_v * log2(e) / T - log2(-log2 u_v))
    // (the two differ by the constant ln(ln 2) and the factor ln 2), i.e. two bare v_log_f32 and one fma per logit
    const float inv_t = 1.0f / e.temp, inv_t_log2e = inv_t * 1.44269504088896340736f;
    // LDS scratch (the GEMM stages are dead after run()'s final barrier): [wn][128 rows][5 words]
    float* red = reinterpret_cast<float*>(smem);
    const int V = p.N;
    // the bias vectors of the lane's columns and the logical row numbers are loaded ONCE, ahead of the row-block loop (inside it each
    // was a dependent L2 round trip per 16x16 block)
    f32x4 bvj[VTN];
#pragma unroll
    for (int j = 0; j < VTN; ++j) {
        const int n = n0 + wn * 16 * VTN + j * 16 + g * 4;
        bvj[j] = n < V ? *reinterpret_cast<const f32x4*>(e.bias + n) : f32x4{0, 0, 0, 0};         // V % 4 == 0 (host check)
    }
    long lrows[VTM];
#pragma unroll
    for (int i = 0; i < VTM; ++i) {
        const int m = m0 + wm * 16 * VTM + i * 16 + lr;
        lrows[i] = m < p.M ? (e.rows ? e.rows[m] : m) : 0;
    }
#pragma unroll
    for (int i = 0; i < VTM; ++i) {
        const int ml = wm * 16 * VTM + i * 16 + lr;
        const int m = m0 + ml;
        const bool mok = m < p.M;
        const long lrow = lrows[i];
        float best = -INFINITY, blog = 0.f, lmax = -INFINITY;
        int bidx = 0x7fffffff;
        float lg[4 * VTN];
#pragma unroll
        for (int j = 0; j < VTN; ++j) {
            const int n = n0 + wn * 16 * VTN + j * 16 + g * 4;
            const f32x4 bv = bvj[j];
            f32x4 uv = f32x4{0.5f, 0.5f, 0.5f, 0.5f};
            if (PARITY) {
                if (mok && n < V) {
                    if (e.U) uv = *reinterpret_cast<const f32x4*>(e.U + (size_t)lrow * V + n);
                    else {
                        const uint64_t li = (uint64_t)lrow * (uint64_t)V + (uint64_t)n;
#pragma unroll
                        for (int r = 0; r < 4; ++r) uv[r] = torch_uniform(e.seed_lo, e.seed_hi, e.philox_offset, li + r, e.philox_stride);
                    }
                }
            }
            float uf[4] = {0.5f, 0.5f, 0.5f, 0.5f};
            if (!PARITY && !e.no_noise) {
                const uint64_t gq = ((uint64_t)lrow * (uint64_t)V + (uint64_t)n) >> 2;      // V % 4 == 0: group of 4 columns
                uniform24x4(seed_lo, seed_hi, (uint32_t)gq, (uint32_t)(gq >> 32), uf);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n + r;
                const float logit = acc[i][j][r] + bv[r];
                float noisy;
                if (PARITY) {
                    const float gum = -logf(-logf(uv[r] + 1e-10f) + 1e-10f);
                    noisy = logit / e.temp + gum;
                } else {
                    noisy = fmaf(logit, inv_t_log2e, -__log2f(-__log2f(uf[r])));           // u = 0: -inf, never chosen
                }
                if (e.no_noise) noisy = logit;
                const bool ok = nn < V;
                if (LSE) lg[j * 4 + r] = ok ? logit : -INFINITY;
                if (ok && (noisy > best)) { best = noisy; bidx = nn; blog = logit; }   // ascending nn: first max wins
                if (LSE && ok) lmax = fmaxf(lmax, logit);
            }
        }
        float lsum = 0.f;
        if (LSE) {
#pragma unroll
            for (int q = 0; q < 4 * VTN; ++q) lsum += __expf(lg[q] - lmax);         // exp(-inf) = 0 for padding
        }
        // combine the 4 lane groups holding the same row
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float ob = __shfl_xor(best, off, 64), ol = __shfl_xor(blog, off, 64);
            const int oi = __shfl_xor(bidx, off, 64);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; blog = ol; }
            if (LSE) {
                const float om = __shfl_xor(lmax, off, 64), os = __shfl_xor(lsum, off, 64);
                const float nm = fmaxf(lmax, om);
                lsum = (lmax == -INFINITY ? 0.f : lsum * __expf(lmax - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
                lmax = nm;
            }
        }
        if (g == 0) {
            float* rr = red + ((size_t)wn * 128 + ml) * 5;
            rr[0] = best; rr[1] = __builtin_bit_cast(float, bidx); rr[2] = blog; rr[3] = lmax; rr[4] = lsum;
        }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int ml = threadIdx.x, m = m0 + ml;
        if (m < p.M) {
            const float* a = red + (size_t)ml * 5;
            float best = a[0], blog = a[2], lmax = a[3], lsum = a[4];
            int bidx = __builtin_bit_cast(int, a[1]);
#pragma unroll
            for (int q = 1; q < VWN; ++q) {
                const float* b = red + ((size_t)q * 128 + ml) * 5;
                const int oi = __builtin_bit_cast(int, b[1]);
                if (b[0] > best || (b[0] == best && oi < bidx)) { best = b[0]; bidx = oi; blog = b[2]; }
                if (LSE) {
                    const float nm = fmaxf(lmax, b[3]);
                    lsum = (lmax == -INFINITY ? 0.f : lsum * __expf(lmax - nm)) + (b[3] == -INFINITY ? 0.f : b[4] * __expf(b[3] - nm));
                    lmax = nm;
                }
            }
            const size_t o = (size_t)ntile * p.M + m;
            e.p_val[o] = best; e.p_idx[o] = bidx; e.p_logit[o] = blog;
            if (LSE) { e.p_max[o] = lmax; e.p_sum[o] = lsum; }
        }
    }
}

// ---- pk_vocab_sample, bf16, D = 512, M >= 1024: the A-RESIDENT persistent form (round 4) ------------------------------------------------
// The tiled kernel above fills BOTH operands of every 128 x 128 tile through the LDS-DMA ring (256 KB per 16.8 MFLOP) and starts a cold
// 8-k-tile pipeline per tile; profiles/gemm_yardstick_r03.txt has it at 668 TF against the vendor GEMM's 842 on this shape, and the
// fill-path benchmark (profiles/fill_path_r04.txt) shows that moving one operand to VGPR loads does not help by itself (same 64 B/clk L1 path).
// What does halve the fill is not re-fetching A at all: with K = 512 a 128-row panel of A is 128 KB -- it FITS the 160 KB LDS.  So:
//   * one workgroup per CU (128 KB of LDS, 8 waves), persistent: worker (xcd, s) owns a contiguous range of (row tile, group of 8 vocabulary
//     tiles) units of its XCD's eighth of the vocabulary; the A panel is loaded once per row tile it visits (1-2 per worker, LDS-DMA);
//   * inside a unit wave w owns vocabulary tile 8 * group + w and walks its 128 columns as 8 sub-blocks of 16.  The 16 x 512 slice of W of a
//     sub-block goes global -> VGPR as the MFMA operand itself (a lane's fragment chunk is 16 contiguous bytes of one W row): 16 loads of 16 B
//     per lane, issued ONE SUB-BLOCK AHEAD into a register ring of 8 k-tiles (64 VGPRs) -- 16 KB per wave / 128 KB per CU in flight, which
//     is what the ~1 us L2 -> CU latency needs at this rate (the first version streamed W through a private 2-slice LDS ring, 4 KB per wave in
//     flight: latency-bound at the tiled kernel's speed, profiles/vocab_resident_r04.txt).  No LDS-DMA, no s_waitcnt by hand and no
//     workgroup barrier in the loop: the waves drift apart and fill the MFMA pipe through each other's waits;
//   * 8 row fragments x 16 columns per wave: 16 MFMAs per k-tile against A fragments read from the resident panel (16 KB per wave per k-tile
//     of the 256 B/clk LDS read path);
//   * the gumbel-argmax / log-sum-exp epilogue runs per sub-block on the lane's 32 logits with a running (best, index, logit, max, sum) per
//     row; after the 8th sub-block the 4 lane groups are folded and the tile's partial is written -- the SAME [tile][row] partials as the tiled
//     kernel, so pk_vocab_reduce / pk_vocab_ce are unchanged.  Noise is indexed by (logical row, column) in both kernels: same draws.
// Fill per 16.8 MFLOP: 128 KB (W only).  The k-tiles of a slice are walked from a per-wave rotation (L2 channel spread, as krot above).
constexpr int VR_NT = 8;
constexpr int VR_SMEM = VR_NT * 128 * 128;                              // the A panel: 8 k-tile blocks of [128 rows][128 B]

// NW waves per workgroup (8: 2 per SIMD, <= 256 VGPRs; 12: 3 per SIMD, <= 168 VGPRs), RING k-tiles of W in flight per lane (8 = one whole
// sub-block ahead, 4 = half of one), FADB: the A fragments of step t + 1 are read before the MFMAs of step t (costs 32 VGPRs), TNW: column
// fragments per sub-block (1: 16 columns; 2: 32 columns -- every A fragment read from LDS then feeds two MFMAs, half the LDS traffic per flop; built and
// measured: 256 VGPRs with spills, 399 - 423 us argmax-only vs 405, 493 - 531 us with the noise vs 424 - 443: not instantiated).
// Work: the (row tile, vocabulary tile) pairs of the XCD's eighth of the vocabulary, row major; worker s takes a contiguous range of them, its
// wave w every NW-th pair of each row tile's part of the range.
template <int NW, int RING, bool FADB, int TNW, bool PARITY, bool LSE>
__global__ __launch_bounds__(64 * NW) void vocab_resident_kernel(const GemmOperands p, const VocabArgs e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr int NT = VR_NT;
    static_assert(RING == 4 || RING == 8, "ring of 4 or 8 k-tiles");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lr = lane & 15;
    const int MT = (p.M + 127) / 128, V = p.N;
    const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3, nworkers = gridDim.x >> 3;
    const int CTX = (e.ntiles + 7) / 8;                                 // vocabulary tiles per XCD range
    const int c_lo = xcd * CTX, c_hi = (c_lo + CTX < e.ntiles) ? c_lo + CTX : e.ntiles;
    const int CT = c_hi - c_lo;                                         // vocabulary tiles of this XCD
    if (CT <= 0) return;
    // Segments of this worker, in order: (1) `full` = MT / nworkers sweeps in which worker s owns row tile j * nworkers + s and walks ALL vocabulary
    // tiles of the XCD's range -- the 32 workers of an XCD then read the same W tile at about the same time, so it is fetched into the XCD's L2
    // once per sweep instead of once per row tile (a contiguous split of the row-major pair list had every worker at a different tile: the 8 MB range
    // cycled through the 4 MB L2); (2) the pairs of the MT % nworkers left-over row tiles, split evenly (contiguous, row tile major).
    const int full = MT / nworkers, R = MT - full * nworkers;
    const int Pl = R * CT;
    const int q0 = (int)((long)s * Pl / nworkers), q1 = (int)((long)(s + 1) * Pl / nworkers);
    const int lrow0 = q0 / CT, nleft = q1 > q0 ? (q1 - 1) / CT - lrow0 + 1 : 0;
    const int nseg = full + nleft;
    if (nseg == 0) return;                                              // the whole workgroup leaves together
    const uint32_t bytesA = (uint32_t)p.M * (uint32_t)p.lda * 2u, bytesW = (uint32_t)V * (uint32_t)p.ldw * 2u;
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, bytesA, 0x00020000);
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, bytesW, 0x00020000);
    const int lrow = lane >> 3, lslot = lane & 7, srcslot = lslot ^ lrow;          // A pieces: 8 rows x 8 slots per 1 KiB, swizzle on the source
    const int rot = (wave + s) & (NT - 1);

    // segment k: row tile r and the vocabulary tiles c_lo + [a, b) of it; this wave takes a + wave, + NW, ...
    auto segment = [&](int k, int& r, int& a, int& b) {
        if (k < full) { r = k * nworkers + s; a = 0; b = CT; return; }
        const int l = lrow0 + (k - full), lo = l * CT;
        r = full * nworkers + l;
        a = (q0 > lo ? q0 : lo) - lo;
        b = (q1 < lo + CT ? q1 : lo + CT) - lo;
    };
    // the tile after (segment k, cc) in this wave's order; k2 >= nseg when there is none
    auto next_tile = [&](int k, int cc, int& k2, int& cc2) {
        int r, a, b;
        segment(k, r, a, b);
        if (cc + NW < b) { k2 = k; cc2 = cc + NW; return; }
        for (k2 = k + 1; k2 < nseg; ++k2) {
            segment(k2, r, a, b);
            if (a + wave < b) { cc2 = a + wave; return; }
        }
        cc2 = 0;
    };
    // byte offset of this lane's fragment row of sub-block sb of vocabulary tile c (out of range: reads as 0)
    constexpr int NSB = 8 / TNW, SBW = 16 * TNW;                        // sub-blocks per vocabulary tile, columns per sub-block
    auto w_base = [&](bool valid, int c, int sb, int j) -> uint32_t {
        const int row = c * 128 + sb * SBW + j * 16 + lr;
        return (valid && row < V) ? (uint32_t)row * (uint32_t)p.ldw * 2u + g * 16 : bytesW;
    };
    auto load_a = [&](int r) {
        const int m0 = r * 128;
        if (wave < 8) {                                                 // 128 pieces of 1 KiB: 16 per wave of the first eight
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int piece = wave * 2 + j;
                    const int gm = m0 + piece * 8 + lrow;
                    const uint32_t off = gm < p.M ? (uint32_t)gm * (uint32_t)p.lda * 2u + srcslot * 16 : bytesA;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(smem + kt * 16384 + piece * 1024), 16, off, kt * 128, 0, 0);
                }
        }
    };

    uint32_t seed_lo = e.seed_lo, seed_hi = e.seed_hi;
    if (!PARITY && e.seed_dev) {
        const unsigned long long sd = (((unsigned long long)e.seed_hi << 32) | e.seed_lo) + *e.seed_dev;
        seed_lo = (uint32_t)sd; seed_hi = (uint32_t)(sd >> 32);
    }
    const float inv_t = 1.0f / e.temp, inv_t_log2e = inv_t * 1.44269504088896340736f;

    // the first tile of this wave and the first RING k-tiles of its first sub-block
    int tk, tcc;
    {
        int r, a, b;
        segment(0, r, a, b);
        if (a + wave < b) { tk = 0; tcc = a + wave; }
        else next_tile(0, b, tk, tcc);                                  // (cc = b: nothing further in this segment)
    }
    u32x4 wreg[RING][TNW][2];
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const uint32_t base = w_base(tk < nseg, c_lo + tcc, 0, j);
#pragma unroll
        for (int kt = 0; kt < RING; ++kt)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
                wreg[kt][j][ch] = __builtin_amdgcn_raw_buffer_load_b128(rsW, base + ch * 64, ((kt + rot) & (NT - 1)) * 128, 0);
    }
    int lrows[8];
    for (int k = 0; k < nseg; ++k) {
        int r, sa, sb_;
        segment(k, r, sa, sb_);
        const int m0 = r * 128;
        // ---- swap the resident panel (every wave takes part, with or without tiles in this segment)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // every wave is done reading the old panel
        load_a(r);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + i * 16 + lr;
            lrows[i] = m < p.M ? (e.rows ? e.rows[m] : m) : 0;
        }
        while (tk == k) {
            const int c = c_lo + tcc;
            int nk, ncc;
            next_tile(tk, tcc, nk, ncc);
            float best[8], blog[8], lmax[8], lsum[8];
            int bidx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; blog[i] = 0.f; bidx[i] = 0x7fffffff; lmax[i] = -INFINITY; lsum[i] = 0.f; }
#pragma unroll 1
            for (int sb = 0; sb < NSB; ++sb) {
                f32x4 bvs[TNW];
                uint32_t cbase[TNW], nbase[TNW];                        // refill targets: later k-tiles of THIS sub-block (RING = 4) / the first RING of the NEXT one
#pragma unroll
                for (int j = 0; j < TNW; ++j) {
                    const int nj = c * 128 + sb * SBW + j * 16 + g * 4;
                    bvs[j] = nj < V ? *reinterpret_cast<const f32x4*>(e.bias + nj) : f32x4{0, 0, 0, 0};
                    cbase[j] = w_base(true, c, sb, j);
                    nbase[j] = sb < NSB - 1 ? w_base(true, c, sb + 1, j) : w_base(nk < nseg, c_lo + ncc, 0, j);
                }
                f32x4 acc[8][TNW];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
                // 16 steps (k-tile, chunk); each consumed ring slot is refilled with the k-tile RING steps ahead
                Frag<bf16> fa[FADB ? 2 : 1][8];
                if (FADB) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) lds_frag(fa[0][i], smem + (rot & (NT - 1)) * 16384, i * 16 + lr, 0, g);
                }
#pragma unroll
                for (int t = 0; t < 2 * NT; ++t) {
                    const int kt = t >> 1, ch = t & 1;
                    if (FADB) {
                        if (t + 1 < 2 * NT) {
                            const int kt1 = (t + 1) >> 1, ch1 = (t + 1) & 1;
                            const char* ablk1 = smem + ((kt1 + rot) & (NT - 1)) * 16384;
#pragma unroll
                            for (int i = 0; i < 8; ++i) lds_frag(fa[(t + 1) & 1][i], ablk1, i * 16 + lr, ch1, g);
                        }
                    } else {
                        const char* ablk = smem + ((kt + rot) & (NT - 1)) * 16384;
#pragma unroll
                        for (int i = 0; i < 8; ++i) lds_frag(fa[0][i], ablk, i * 16 + lr, ch, g);
                    }
                    const int fkt = kt + RING;                          // the k-tile that takes this ring slot
#pragma unroll
                    for (int j = 0; j < TNW; ++j) {
                        Frag<bf16> fw;
                        fw.v = wreg[kt % RING][j][ch];
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i][j] = mma(fw, fa[FADB ? (t & 1) : 0][i], acc[i][j]);
                        wreg[kt % RING][j][ch] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (fkt < NT ? cbase[j] : nbase[j]) + ch * 64,
                                                                                      (((fkt & (NT - 1)) + rot) & (NT - 1)) * 128, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);                  // keep this order (an unconstrained schedule hoisted every read and spilled)
                }
                // ---- epilogue of the sub-block: 8 row fragments x TNW groups of 4 consecutive columns per lane
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = m0 + i * 16 + lr;
                    const bool mok = m < p.M;
                    const long lrow_ = lrows[i];
#pragma unroll
                    for (int j = 0; j < TNW; ++j) {
                        const int n = c * 128 + sb * SBW + j * 16 + g * 4;
                        f32x4 uv = f32x4{0.5f, 0.5f, 0.5f, 0.5f};
                        if (PARITY) {
                            if (mok && n < V) {
                                if (e.U) uv = *reinterpret_cast<const f32x4*>(e.U + (size_t)lrow_ * V + n);
                                else {
                                    const uint64_t li = (uint64_t)lrow_ * (uint64_t)V + (uint64_t)n;
#pragma unroll
                                    for (int rr = 0; rr < 4; ++rr) uv[rr] = torch_uniform(e.seed_lo, e.seed_hi, e.philox_offset, li + rr, e.philox_stride);
                                }
                            }
                        }
                        float uf[4] = {0.5f, 0.5f, 0.5f, 0.5f};
                        if (!PARITY && !e.no_noise) {
                            const uint64_t gq = ((uint64_t)lrow_ * (uint64_t)V + (uint64_t)n) >> 2;
                            uniform24x4(seed_lo, seed_hi, (uint32_t)gq, (uint32_t)(gq >> 32), uf);
                        }
                        float lg[4];
                        float bmax = -INFINITY;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int nn = n + rr;
                            const float logit = acc[i][j][rr] + bvs[j][rr];
                            float noisy;
                            if (PARITY) {
                                const float gum = -logf(-logf(uv[rr] + 1e-10f) + 1e-10f);
                                noisy = logit / e.temp + gum;
                            } else {
                                noisy = fmaf(logit, inv_t_log2e, -__log2f(-__log2f(uf[rr])));
                            }
                            if (e.no_noise) noisy = logit;
                            const bool ok = nn < V;
                            if (LSE) { lg[rr] = ok ? logit : -INFINITY; bmax = fmaxf(bmax, lg[rr]); }
                            if (ok && (noisy > best[i])) { best[i] = noisy; bidx[i] = nn; blog[i] = logit; }      // ascending nn: first max wins
                        }
                        if (LSE && bmax > -INFINITY) {
                            const float nm = fmaxf(lmax[i], bmax);
                            float add = 0.f;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) add += __expf(lg[rr] - nm);
                            lsum[i] = (lmax[i] == -INFINITY ? 0.f : lsum[i] * __expf(lmax[i] - nm)) + add;
                            lmax[i] = nm;
                        }
                    }
                }
            }
            // ---- the tile is done: fold the 4 lane groups of every row and write the [tile][row] partial
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float b_ = best[i], l_ = blog[i], mx = lmax[i], sm = lsum[i];
                int ix = bidx[i];
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const float ob = __shfl_xor(b_, off, 64), ol = __shfl_xor(l_, off, 64);
                    const int oi = __shfl_xor(ix, off, 64);
                    if (ob > b_ || (ob == b_ && oi < ix)) { b_ = ob; ix = oi; l_ = ol; }
                    if (LSE) {
                        const float om = __shfl_xor(mx, off, 64), os = __shfl_xor(sm, off, 64);
                        const float nm = fmaxf(mx, om);
                        sm = (mx == -INFINITY ? 0.f : sm * __expf(mx - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
                        mx = nm;
                    }
                }
                const int m = m0 + i * 16 + lr;
                if (g == 0 && m < p.M) {
                    const size_t o = (size_t)c * p.M + m;
                    e.p_val[o] = b_; e.p_idx[o] = ix; e.p_logit[o] = l_;
                    if (LSE) { e.p_max[o] = mx; e.p_sum[o] = sm; }
                }
            }
            tk = nk; tcc = ncc;
        }
    }
}

// fold the per-tile partials: a workgroup owns 32 consecutive rows (coalesced 128-byte reads of the [tile][row]
// partial arrays), its 8 thread groups stride over the tiles and meet in LDS; then the (B, n) state is updated,
// optionally scattered through rows[]
__global__ __launch_bounds__(256) void vocab_reduce_kernel(const float* __restrict__ p_val, const int* __restrict__ p_idx,
                                                           const float* __restrict__ p_logit, const float* __restrict__ p_max,
                                                           const float* __restrict__ p_sum, int ntiles, int M,
                                                           const int* __restrict__ rows, const unsigned char* __restrict__ mask,
                                                           long long* __restrict__ ids, long long* __restrict__ pred,
                                                           float* __restrict__ scores, int need_lse) {
    __shared__ float red[8][32][5];
    const int rr = threadIdx.x & 31, tg = threadIdx.x >> 5;
    const int m = blockIdx.x * 32 + rr;
    float best = -INFINITY, blog = 0.f, lmax = -INFINITY, lsum = 0.f;
    int bidx = 0x7fffffff;
    if (m < M) {
        // 8 tiles' partials are requested together and folded in tile order: one load -> compare chain per tile was 64 dependent
        // round trips per thread (the kernel ran at 1.3 TB/s on 28 MB)
        for (int t0 = tg; t0 < ntiles; t0 += 64) {
            float v[8], lg[8], om[8], os[8];
            int ix[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 8;
                const size_t o = (size_t)(t < ntiles ? t : tg) * M + m;
                v[u] = p_val[o]; ix[u] = p_idx[o]; lg[u] = p_logit[o];
                if (need_lse) { om[u] = p_max[o]; os[u] = p_sum[o]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u * 8 >= ntiles) break;
                if (v[u] > best || (v[u] == best && ix[u] < bidx)) { best = v[u]; bidx = ix[u]; blog = lg[u]; }
                if (need_lse) {
                    const float nm = fmaxf(lmax, om[u]);
                    lsum = (lmax == -INFINITY ? 0.f : lsum * __expf(lmax - nm)) + (om[u] == -INFINITY ? 0.f : os[u] * __expf(om[u] - nm));
                    lmax = nm;
                }
            }
        }
    }
    red[tg][rr][0] = best; red[tg][rr][1] = __builtin_bit_cast(float, bidx); red[tg][rr][2] = blog;
    red[tg][rr][3] = lmax; red[tg][rr][4] = lsum;
    __syncthreads();
    if (tg != 0 || m >= M) return;
    for (int q = 1; q < 8; ++q) {
        const float v = red[q][rr][0];
        const int ix = __builtin_bit_cast(int, red[q][rr][1]);
        if (v > best || (v == best && ix < bidx)) { best = v; bidx = ix; blog = red[q][rr][2]; }
        if (need_lse) {
            const float om = red[q][rr][3], os = red[q][rr][4];
            const float nm = fmaxf(lmax, om);
            lsum = (lmax == -INFINITY ? 0.f : lsum * __expf(lmax - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
            lmax = nm;
        }
    }
    const int r = rows ? rows[m] : m;
    if (pred) pred[r] = bidx;
    const bool mk = mask ? mask[r] != 0 : true;
    if (ids && mk) ids[r] = bidx;
    if (scores && need_lse) scores[r] = mk ? 1.0f - __expf(blog - lmax) / lsum : -1e4f;
}

// cross entropy of the vocab head WITHOUT the logits (the masked-token objective, phenaki_pytorch.py:640-643): one wave per
// row folds the (max, sum exp) partials pk_vocab_sample left per 128-column tile into lse, and takes the target logit as
// the dot product of the row with ONE row of W (+ bias): loss[m] = lse - logit[target]
// element c of a weight row as the main loop sees it (split-bf16: hi + lo of the host-packed planes, block c / 32)
template <typename T> __device__ __forceinline__ float w_elem(const T* wrow, int c) { return load_elem(wrow + c); }
template <> __device__ __forceinline__ float w_elem<bf16x3>(const bf16x3* wrow, int c) {
    const u16* blk = reinterpret_cast<const u16*>(wrow + (c & ~31));
    return bf2f(blk[c & 31]) + bf2f(blk[32 + (c & 31)]);
}

template <typename T>
__global__ __launch_bounds__(256) void vocab_ce_kernel(const float* __restrict__ p_max, const float* __restrict__ p_sum, int ntiles, int M,
                                                       const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias, int D, const long long* __restrict__ targets,
                                                       const int* __restrict__ rows, float* __restrict__ loss, float* __restrict__ lse_out) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float lmax = -INFINITY, lsum = 0.f;
    for (int t = lane; t < ntiles; t += 64) {
        const size_t o = (size_t)t * M + m;
        const float om = p_max[o], os = p_sum[o];
        const float nm = fmaxf(lmax, om);
        lsum = (lmax == -INFINITY ? 0.f : lsum * __expf(lmax - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
        lmax = nm;
    }
    const float gmax = wave_max(lmax);
    const float gsum = wave_sum(lmax == -INFINITY ? 0.f : lsum * __expf(lmax - gmax));
    const long long tgt = targets[rows ? rows[m] : m];
    const T* a = A + (size_t)m * lda;
    const T* w = W + (size_t)tgt * ldw;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot += load_elem(a + c) * w_elem<T>(w, c);
    dot = wave_sum(dot);
    if (lane == 0) {
        const float lse = gmax + logf(gsum);
        loss[m] = lse - (dot + bias[tgt]);
        if (lse_out) lse_out[m] = lse;
    }
}

// ---- backward of that cross entropy, one SLAB of vocabulary columns at a time (the first training kernel, SURVEY.md 8f row 1;
// reference phenaki_pytorch.py:640-643 under autograd): with loss = mean_m (lse_m - logit[m][t_m]) over the M rows,
//     g[m][v] = d loss / d logit[m][v] = (exp(logit[m][v] - lse_m) - [v == t_m]) * scale,      scale = upstream gradient / M,
// and dE = g W, dW = g^T E, db = colsum(g).  The (M, V) logits / g matrices never exist: the host loops over slabs of Vs columns,
// recomputes the slab's logits with pk_gemm, turns them into g IN PLACE here -- also written transposed, the A operand of the dW GEMM --
// and feeds both to pk_gemm (dE accumulates through the residual input).  Tile: 32 x 32 through LDS (coalesced both ways).
// TO = float (exact-f32 / split-bf16 GEMMs read f32 A operands) or bf16.
template <typename TO>
__global__ __launch_bounds__(256) void ce_grad_slab_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ lse,
                                                           const long long* __restrict__ targets, const int* __restrict__ rows, int M, int Vs,
                                                           int v0, float scale_, const float* __restrict__ scale_dev, TO* __restrict__ g, int ldg,
                                                           TO* __restrict__ gT, int ldgt) {
    __shared__ float tile[32][33];
    const float scale = scale_dev ? scale_ * scale_dev[0] : scale_;       // the upstream gradient stays on the device (no host read in backward)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8 threads, 4 rows each
    const int mb = blockIdx.y * 32, vb = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = ty + i * 8, m = mb + ml, v = vb + tx;
        float x = 0.f;
        if (m < M && v < Vs) {
            const long long t = targets[rows ? rows[m] : m];
            x = (__expf(logits[(size_t)m * ldl + v] - lse[m]) - (t == (long long)(v0 + v) ? 1.f : 0.f)) * scale;
            store_elem(g + (size_t)m * ldg + v, x);
        }
        tile[ml][tx] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int vl = ty + i * 8, v = vb + vl, m = mb + tx;
        if (v < Vs && m < ldgt) store_elem(gT + (size_t)v * ldgt + m, m < M ? tile[tx][vl] : 0.f);      // pad columns [M, ldgt) are zeroed
    }
}

// out[r] = sum_c x[r][c] (one wave per row; fixed order: deterministic) -- db of a slab = row sums of g^T
template <typename TI>
__global__ __launch_bounds__(256) void rowsum_kernel(const TI* __restrict__ x, int ldx, int R, int C, float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += load_elem(x + (size_t)r * ldx + c);
    s = wave_sum(s);
    if (lane == 0) out[r] = s;
}

// mask = top-k of scores per row (ties: lower index first), ids = where(mask, mask_id, ids)
__global__ __launch_bounds__(256) void topk_mask_kernel(const float* __restrict__ scores, int n, int k, long long mask_id,
                                                        unsigned char* __restrict__ mask, long long* __restrict__ ids,
                                                        int* __restrict__ rows_out, float* __restrict__ scores_next) {
    // grid (B, ceil(n / 64)): a workgroup ranks 64 positions of one batch row, 4 lanes per position each scanning a
    // quarter of the row from LDS (B workgroups of 3 x n serial compares each took 47 us at B = 8, n = 576)
    extern __shared__ float sc[];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += 256) sc[i] = scores[(size_t)b * n + i];
    __syncthreads();
    const int i = blockIdx.y * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const int q = (n + 3) >> 2;
    const int j0 = part * q, j1 = j0 + q < n ? j0 + q : n;
    const float v = i < n ? sc[i] : 0.f;
    int rank = 0;
    for (int j = j0; j < j1; ++j) {
        const float w = sc[j];
        rank += (w > v || (w == v && j < i)) ? 1 : 0;
    }
    rank += __shfl_xor(rank, 1, 64);
    rank += __shfl_xor(rank, 2, 64);
    if (part != 0 || i >= n) return;
    // critic-less sampling: the next scores are where(mask, 1 - p, -1e4) (phenaki_pytorch.py:547-550); pk_vocab_reduce only
    // visits the masked rows, so the -1e4 of every other position is laid down here, in the OTHER score buffer
    if (scores_next) scores_next[(size_t)b * n + i] = -1e4f;
    const bool sel = rank < k;
    mask[(size_t)b * n + i] = sel ? 1 : 0;
    if (sel) ids[(size_t)b * n + i] = mask_id;
    if (sel && rows_out) rows_out[(size_t)b * k + rank] = b * n + i;         // compacted list of the masked positions
}

}  // namespace pk
using namespace pk;

static inline int nblocks(long total) { long b = (total + 255) / 256; return (int)(b < 16384 ? (b > 0 ? b : 1) : 16384); }
#define STREAM(s) reinterpret_cast<hipStream_t>(s)
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int pk_cfg_mix(const float* x, int ldx, int nb, int n_tot, int n_prime, const int* rows, int nrows,
                          float scale, int has_null, void* out, int ldo, int out_is_f32, int D, void* stream) {
    if (!x || !out || nb <= 0 || n_tot <= n_prime || n_prime < 0 || nrows <= 0 || D <= 0) return PK_EINVAL;
    if ((D & 3) || (ldx & 3) || (ldo & 3)) return PK_EALIGN;
    hipLaunchKernelGGL(cfg_mix_kernel, dim3(nblocks((long)nrows * (D >> 2))), dim3(256), 0, STREAM(stream),
                       x, ldx, nb, n_tot, n_prime, rows, nrows, scale, has_null, out, ldo, out_is_f32, D);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_vocab_ntiles(int V) { return (V + 127) / 128; }

// workspace: 5 arrays of ntiles*M 4-byte words, passed as one buffer `partials` of 5*ntiles*M words
static int vocab_sample_launch(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias,
                               int M, int V, int D, float temperature, const float* U, const int* rows,
                               unsigned long long seed, const unsigned long long* seed_dev, int need_lse, void* partials, void* stream,
                               unsigned long long philox_offset, unsigned int philox_stride) {
    if (!A || !W || !bias || !partials || M <= 0 || V <= 0 || D <= 0) return PK_EINVAL;
    if (dtype != 0 && dtype != 1 && dtype != 2) return PK_EINVAL;
    const int eps = dtype == 1 ? 8 : 4;
    if ((V & 3) || (D % eps) || (lda % eps) || (ldw % eps) || !al16(A) || !al16(W) || !al16(bias) || (U && !al16(U))) return PK_EALIGN;
    const int ntiles = pk_vocab_ntiles(V);
    const int bk = dtype == 1 ? 64 : 32;                  // LDS-DMA main loop: W zero-padded along K to the k-tile
    if (ldw < (D + bk - 1) / bk * bk) return PK_EINVAL;
    if ((size_t)M * lda * (dtype == 1 ? 2 : 4) >= 0xFFFFFFF0ull || (size_t)V * ldw * (dtype == 1 ? 2 : 4) >= 0xFFFFFFF0ull) return PK_EINVAL;
    GemmOperands p{A, W, nullptr, lda, ldw, M, V, D, 0, krot_default()};
    VocabArgs e;
    e.bias = bias; e.U = U; e.rows = rows;
    e.philox_stride = philox_stride; e.philox_offset = philox_offset;
    const bool parity = U != nullptr || philox_stride != 0;
    e.temp = temperature > 1e-10f ? temperature : 1e-10f;
    e.seed_lo = (uint32_t)seed; e.seed_hi = (uint32_t)(seed >> 32); e.seed_dev = seed_dev;
    e.need_lse = need_lse & 1; e.no_noise = (need_lse >> 1) & 1; e.ntiles = ntiles;
    static const int panel_env = [] { const char* v = getenv("PK_VOCAB_PANEL"); return v ? atoi(v) : 1; }();
    e.plain_order = panel_env == 0;
    const size_t sz = (size_t)ntiles * M;
    e.p_val = reinterpret_cast<float*>(partials);
    e.p_idx = reinterpret_cast<int*>(partials) + sz;
    e.p_logit = reinterpret_cast<float*>(partials) + 2 * sz;
    e.p_max = reinterpret_cast<float*>(partials) + 3 * sz;
    e.p_sum = reinterpret_cast<float*>(partials) + 4 * sz;
    dim3 grid(8 * ((ntiles + 7) / 8) * ((M + 127) / 128));        // 1-D: see the tile order in the kernel
    hipStream_t s = STREAM(stream);
    // bf16, K = 512, enough rows for every CU: the A-resident persistent kernel.  Measured on 4608 x 65536 x 512 (profiles/vocab_resident_r04.txt):
    // twelve waves with a ring of 4 k-tiles (3 waves per SIMD, 153 VGPRs) win when the epilogue carries the noise hash (432 vs 451 us), eight
    // waves with a ring of 8 and double-buffered A fragments (2 per SIMD) when it carries the log-sum-exp state (the 12-wave build of that
    // variant spills); below ~2048 rows the panel swap and the start-up are no longer amortised (1152 rows: 146 vs 139 us tiled).
    // PK_VOCAB_RESIDENT: 0 = the tiled kernel, 1 / 2 = force the 8- / 12-wave build, unset = the rule above.
    static const int resident_env = [] { const char* v = getenv("PK_VOCAB_RESIDENT"); return v ? atoi(v) : 3; }();
    if (dtype == 1 && resident_env && D == 64 * VR_NT && M >= (resident_env == 3 ? 2048 : 1024)) {
        const bool lse = e.need_lse != 0;
        const int vi = (parity ? 2 : 0) + (lse ? 1 : 0);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PK_ELAUNCH;
        const dim3 rgrid(256);                                         // 8 XCDs x 32 CUs: one workgroup per CU
#define PK_VR(NWV, RNG, FA, TW, PAR, LS) do { \
            static bool attr_set[64] = {}; \
            if (!attr_set[dev]) { \
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&vocab_resident_kernel<NWV, RNG, FA, TW, PAR, LS>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, VR_SMEM) != hipSuccess) return PK_ELAUNCH; \
                attr_set[dev] = true; \
            } \
            hipLaunchKernelGGL((vocab_resident_kernel<NWV, RNG, FA, TW, PAR, LS>), rgrid, dim3(64 * NWV), VR_SMEM, s, p, e); } while (0)
        if (resident_env == 2 || (resident_env == 3 && !lse)) {
            switch (vi) { case 0: PK_VR(12, 4, false, 1, false, false); break; case 1: PK_VR(12, 4, false, 1, false, true); break;
                          case 2: PK_VR(12, 4, false, 1, true, false); break; default: PK_VR(12, 4, false, 1, true, true); break; }
        } else {
            switch (vi) { case 0: PK_VR(8, 8, true, 1, false, false); break; case 1: PK_VR(8, 8, true, 1, false, true); break;
                          case 2: PK_VR(8, 8, true, 1, true, false); break; default: PK_VR(8, 8, true, 1, true, true); break; }
        }
#undef PK_VR
        PK_CHECK_LAUNCH();
        return PK_OK;
    }
#define PK_VS(TT, PAR, LS) hipLaunchKernelGGL((vocab_sample_kernel<TT, PAR, LS>), grid, dim3(VocabTile<TT>::THREADS), VocabTile<TT>::SMEM, s, p, e)
    const bool lse = e.need_lse != 0;
    if (dtype == 1) {
        if (parity) { if (lse) PK_VS(bf16, true, true); else PK_VS(bf16, true, false); }
        else { if (lse) PK_VS(bf16, false, true); else PK_VS(bf16, false, false); }
    } else if (dtype == 2) {                              // split-bf16: f32 rows of A, host-packed (hi | lo) planes of W (common.hpp)
        if (parity) { if (lse) PK_VS(bf16x3, true, true); else PK_VS(bf16x3, true, false); }
        else { if (lse) PK_VS(bf16x3, false, true); else PK_VS(bf16x3, false, false); }
    } else {
        if (parity) { if (lse) PK_VS(float, true, true); else PK_VS(float, true, false); }
        else { if (lse) PK_VS(float, false, true); else PK_VS(float, false, false); }
    }
#undef PK_VS
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_vocab_sample(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias,
                               int M, int V, int D, float temperature, const float* U, const int* rows,
                               unsigned long long seed, const unsigned long long* seed_dev, int need_lse, void* partials, void* stream) {
    return vocab_sample_launch(dtype, A, lda, W, ldw, bias, M, V, D, temperature, U, rows, seed, seed_dev, need_lse, partials, stream, 0ull, 0u);
}

// PARITY sampling on torch's own device RNG stream: the gumbel noise of logical row r, column v is what
// `torch.zeros(rows_total, V, device=...).uniform_(0, 1)[r][v]` would hold for a generator at (torch_seed, philox_offset) -- the reference's
// gumbel_noise(logits) (phenaki_pytorch.py:88-93) -- generated in the epilogue, the (rows_total, V) noise tensor never exists.  philox_stride =
// 256 * min(#CUs * (max threads per CU / 256), ceil(rows_total * V / 256)); the caller advances the generator (common.hpp torch_uniform).
extern "C" int pk_vocab_sample_philox(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, int M, int V, int D, float temperature,
                                      const int* rows, unsigned long long torch_seed, unsigned long long philox_offset, unsigned int philox_stride,
                                      int need_lse, void* partials, void* stream) {
    if (philox_stride == 0 || (philox_stride & 255) || (philox_offset & 3)) return PK_EINVAL;
    return vocab_sample_launch(dtype, A, lda, W, ldw, bias, M, V, D, temperature, nullptr, rows, torch_seed, nullptr, need_lse & 1, partials, stream,
                               philox_offset, philox_stride);
}

extern "C" int pk_vocab_reduce(const void* partials, int M, int V, const int* rows, const unsigned char* mask,
                               long long* ids, long long* pred, float* scores, int need_lse, void* stream) {
    if (!partials || M <= 0 || V <= 0) return PK_EINVAL;
    const int ntiles = pk_vocab_ntiles(V);
    const size_t sz = (size_t)ntiles * M;
    const float* f = reinterpret_cast<const float*>(partials);
    hipLaunchKernelGGL(vocab_reduce_kernel, dim3((M + 31) / 32), dim3(256), 0, STREAM(stream),
                       f, reinterpret_cast<const int*>(partials) + sz, f + 2 * sz, f + 3 * sz, f + 4 * sz, ntiles, M,
                       rows, mask, ids, pred, scores, need_lse);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// loss[m] = logsumexp_v(logits[m][v]) - logits[m][targets[r]], r = rows ? rows[m] : m, from the partials of a pk_vocab_sample
// call made with need_lse bit 0 set on the SAME A / W / bias (phenaki_pytorch.py:640-643 F.cross_entropy, reduction left to
// the caller).  targets must be < V.
extern "C" int pk_ce_grad_slab(int out_bf16, const float* logits, int ldl, const float* lse, const long long* targets, const int* rows,
                               int M, int Vs, int v0, float scale, const float* scale_dev, void* g, int ldg, void* gT, int ldgt, float* db, void* stream) {
    if (!logits || !lse || !targets || !g || !gT || M <= 0 || Vs <= 0 || v0 < 0 || ldl < Vs || ldg < Vs || ldgt < M) return PK_EINVAL;
    dim3 grid((Vs + 31) / 32, (ldgt + 31) / 32);
    hipStream_t s = STREAM(stream);
    if (out_bf16) {
        hipLaunchKernelGGL((ce_grad_slab_kernel<bf16>), grid, dim3(256), 0, s, logits, ldl, lse, targets, rows, M, Vs, v0, scale, scale_dev,
                           reinterpret_cast<bf16*>(g), ldg, reinterpret_cast<bf16*>(gT), ldgt);
        if (db) hipLaunchKernelGGL((rowsum_kernel<bf16>), dim3((Vs + 3) / 4), dim3(256), 0, s, reinterpret_cast<const bf16*>(gT), ldgt, Vs, M, db);
    } else {
        hipLaunchKernelGGL((ce_grad_slab_kernel<float>), grid, dim3(256), 0, s, logits, ldl, lse, targets, rows, M, Vs, v0, scale, scale_dev,
                           reinterpret_cast<float*>(g), ldg, reinterpret_cast<float*>(gT), ldgt);
        if (db) hipLaunchKernelGGL((rowsum_kernel<float>), dim3((Vs + 3) / 4), dim3(256), 0, s, reinterpret_cast<const float*>(gT), ldgt, Vs, M, db);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_vocab_ce(int dtype, const void* partials, int M, int V, const void* A, int lda, const void* W, int ldw,
                           const float* bias, int D, const long long* targets, const int* rows, float* loss, float* lse_out, void* stream) {
    if (!partials || !A || !W || !bias || !targets || !loss || M <= 0 || V <= 0 || D <= 0) return PK_EINVAL;
    if (dtype != 0 && dtype != 1 && dtype != 2) return PK_EINVAL;
    const int ntiles = pk_vocab_ntiles(V);
    const size_t sz = (size_t)ntiles * M;
    const float* f = reinterpret_cast<const float*>(partials);
    const dim3 grid((M + 3) / 4);
    if (dtype == 2) {
        hipLaunchKernelGGL((vocab_ce_kernel<bf16x3>), grid, dim3(256), 0, STREAM(stream), f + 3 * sz, f + 4 * sz, ntiles, M,
                           reinterpret_cast<const bf16x3*>(A), lda, reinterpret_cast<const bf16x3*>(W), ldw, bias, D, targets, rows, loss, lse_out);
        PK_CHECK_LAUNCH();
        return PK_OK;
    }
    if (dtype == 1) hipLaunchKernelGGL((vocab_ce_kernel<bf16>), grid, dim3(256), 0, STREAM(stream), f + 3 * sz, f + 4 * sz, ntiles, M,
                                       reinterpret_cast<const bf16*>(A), lda, reinterpret_cast<const bf16*>(W), ldw, bias, D, targets, rows, loss, lse_out);
    else hipLaunchKernelGGL((vocab_ce_kernel<float>), grid, dim3(256), 0, STREAM(stream), f + 3 * sz, f + 4 * sz, ntiles, M,
                            reinterpret_cast<const float*>(A), lda, reinterpret_cast<const float*>(W), ldw, bias, D, targets, rows, loss, lse_out);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_topk_mask(const float* scores, int B, int n, int k, long long mask_id, unsigned char* mask,
                            long long* ids, int* rows_out, float* scores_next, void* stream) {
    if (!scores || !mask || !ids || B <= 0 || n <= 0 || k < 0 || n > 12288 || scores_next == scores) return PK_EINVAL;
    hipLaunchKernelGGL(topk_mask_kernel, dim3(B, (n + 63) / 64), dim3(256), n * sizeof(float), STREAM(stream), scores, n, k, mask_id, mask, ids, rows_out, scores_next);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
