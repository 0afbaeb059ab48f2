// LDS-tiled MFMA GEMM main loop shared by every projection on the Phenaki hot path:
//   acc[m][n] = sum_k A[m][k] * W[n][k]        (W is an nn.Linear weight, row-major [N][K])
//
// Tile: BM = 32*TM rows x BN = 32*TN cols per 256-thread workgroup (4 waves as 2 x 2), k-tile of
// 128 bytes per row (64 bf16 / 32 f32).  Staging is global -> registers -> LDS with the next
// k-tile's global loads issued before the current tile's MFMAs (one barrier per k-tile, two LDS
// stages).  LDS rows are 128 B = 8 slots of 16 B, slot index XOR (row & 7): ds_write_b128 (8-lane
// groups = one row) and ds_read_b128 (the gfx950 16-lane groups) are both conflict-free for bf16.
//
// The product is computed transposed (the weight tile is the MFMA "A" operand), so that lane l ends
// up holding 4 CONSECUTIVE output columns of one output row:
//   acc[i][j][r] = C[m0 + wm*16*TM + i*16 + (l & 15)][n0 + wn*16*TN + j*16 + (l >> 4)*4 + r]
// which makes the epilogue's bias/residual loads and stores 16-byte vectors and puts a GEGLU
// (value, gate) pair in one lane.
#pragma once
#include <cstdlib>
#include "common.hpp"

namespace pk {

// host: k-rotation default (PK_GEMM_KROT=0 switches it off for A/B measurements)
static inline int krot_default() {
    static const int v = getenv("PK_GEMM_KROT") ? atoi(getenv("PK_GEMM_KROT")) : 1;
    return v;
}

struct GemmOperands {
    const void* A;      // [M][lda]   f32 or T
    const void* W;      // [N][ldw]   T
    const int* a_rows;  // optional gather: logical row m reads physical row a_rows[m]
    int lda, ldw;
    int M, N, K;
    int plain_map;      // 1: row-major tile order (A/B benchmarking only); 0: XCD-aware tile map (DMA kernels)
    int krot;           // 1: workgroup b walks the k-tiles starting at tile (b >> 3) % nt (DMA kernels; see gemm_dma.hpp)
    int w_gap_from, w_gap_rows;   // DMA kernels: tile rows >= w_gap_from of W read global row n0 + row + w_gap_rows (two row ranges, one tile)
    int panel;          // DMA kernels, XCD-aware map: m-tiles per L2 panel (0: the XCD's whole chunk is one panel), see xcd_panel_tile
};

// XCD-aware tile order of the LDS-DMA GEMMs.  Workgroup b is observed to run on XCD b % 8 (speed only, never correctness); each XCD
// has a private 4 MiB L2.  XCD x owns a contiguous chunk of `mcount` m-tiles (cmax = the largest chunk) and walks the n-tiles for it,
// m fastest.  For large M the chunk's slice of A no longer fits the L2 (M = 36 864, K = 512 bf16: 4.7 MB per XCD -- the 128-row GEMMs
// fell from 612 to 471 TFLOP/s between M = 18 432 and 36 864, profiles/gemm_yardstick_r03.txt), so the chunk is cut into PANELS of
// `panel` m-tiles (~1 MB of A): a panel walks all n-tiles before the next panel starts, A stays L2-resident, W streams once per panel.
// idx = position of the workgroup inside its XCD (blockIdx.x >> 3), NT = number of n-tiles.  Returns false: no tile (chunk tail).
__device__ __forceinline__ bool xcd_panel_tile(int idx, int cmax, int mcount, int NT, int panel, int& ml, int& ntile) {
    if (panel <= 0 || panel >= cmax) { ml = idx % cmax; ntile = idx / cmax; }
    else {
        const int full = cmax / panel;
        int pi = idx / (panel * NT);
        if (pi > full) pi = full;
        const int base = pi * panel;
        const int pl = pi < full ? panel : cmax - base;         // rows of this (possibly partial, last) panel
        const int r = idx - pi * panel * NT;
        ml = base + r % pl;
        ntile = r / pl;
    }
    return ml < mcount;
}
// host: m-tiles per panel for a BM-row tile of an A with K elements of `esz` bytes per row (about 1 MB of A per panel, at least 4 tiles)
static inline int xcd_panel_rows(int BM, int K, int esz) {
    const long tile_bytes = (long)BM * K * esz;
    long pnl = (1 << 20) / (tile_bytes > 0 ? tile_bytes : 1);
    return (int)(pnl < 4 ? 4 : pnl);
}

template <typename T, typename TA> struct RawSlot;                    // one thread's 16-B LDS slot, pre-conversion
template <typename T> struct RawSlot<T, T> { u32x4 v; };
template <> struct RawSlot<bf16, float> { f32x4 a, b; };

template <typename T>
__device__ __forceinline__ void raw_load(RawSlot<T, T>& r, const T* p, bool ok) {
    r.v = ok ? *reinterpret_cast<const u32x4*>(p) : u32x4{0, 0, 0, 0};
}
__device__ __forceinline__ void raw_load(RawSlot<bf16, float>& r, const float* p, bool ok) {
    if (ok) { r.a = *reinterpret_cast<const f32x4*>(p); r.b = *reinterpret_cast<const f32x4*>(p + 4); }
    else { r.a = f32x4{0, 0, 0, 0}; r.b = r.a; }
}
template <typename T>
__device__ __forceinline__ u32x4 raw_pack(const RawSlot<T, T>& r) { return r.v; }
__device__ __forceinline__ u32x4 raw_pack(const RawSlot<bf16, float>& r) {
    return u32x4{pack_bf2(r.a[0], r.a[1]), pack_bf2(r.a[2], r.a[3]), pack_bf2(r.b[0], r.b[1]), pack_bf2(r.b[2], r.b[3])};
}

__device__ __forceinline__ void lds_frag(Frag<bf16>& f, const char* tile, int row, int chunk, int g) {
    const int slot = (chunk * 4 + g) ^ (row & 7);
    f.v = *reinterpret_cast<const u32x4*>(tile + row * 128 + (slot << 4));
}
__device__ __forceinline__ void lds_frag(Frag<float>& f, const char* tile, int row, int /*chunk*/, int g) {
    const int s0 = (2 * g) ^ (row & 7), s1 = (2 * g + 1) ^ (row & 7);
    f.lo = *reinterpret_cast<const f32x4*>(tile + row * 128 + (s0 << 4));
    f.hi = *reinterpret_cast<const f32x4*>(tile + row * 128 + (s1 << 4));
}

// A-tile / W-tile fragment reads: the same image for bf16 and f32; for the split-bf16 operand type (common.hpp) the A tile holds
// f32 rows (32 k per 128-byte row), split into (hi, lo) planes in registers, and the W tile holds the host-packed planes
// [hi x 32 | lo x 32]: k = g*8 .. g*8+7 of a row is slot g (hi) and slot 4 + g (lo).  Both use the slot ^ (row & 7) swizzle.
template <typename T> __device__ __forceinline__ void lds_frag_a(Frag<T>& f, const char* tile, int row, int chunk, int g) { lds_frag(f, tile, row, chunk, g); }
template <typename T> __device__ __forceinline__ void lds_frag_w(Frag<T>& f, const char* tile, int row, int chunk, int g) { lds_frag(f, tile, row, chunk, g); }
template <> __device__ __forceinline__ void lds_frag_a<bf16x3>(Frag<bf16x3>& f, const char* tile, int row, int /*chunk*/, int g) {
    const int s0 = (2 * g) ^ (row & 7), s1 = (2 * g + 1) ^ (row & 7);
    const f32x4 a = *reinterpret_cast<const f32x4*>(tile + row * 128 + (s0 << 4));
    const f32x4 b = *reinterpret_cast<const f32x4*>(tile + row * 128 + (s1 << 4));
    split8(a, b, f.hi, f.lo);
}
template <> __device__ __forceinline__ void lds_frag_w<bf16x3>(Frag<bf16x3>& f, const char* tile, int row, int /*chunk*/, int g) {
    const int s0 = g ^ (row & 7), s1 = (4 + g) ^ (row & 7);
    f.hi = *reinterpret_cast<const u32x4*>(tile + row * 128 + (s0 << 4));
    f.lo = *reinterpret_cast<const u32x4*>(tile + row * 128 + (s1 << 4));
}

template <typename T, typename TA, int TM, int TN>
struct GemmTile {
    static constexpr int BM = 32 * TM, BN = 32 * TN;
    static constexpr int EPS = 16 / (int)sizeof(T);      // elements per 16-B slot
    static constexpr int BK = 8 * EPS;                   // elements per k-tile
    static constexpr int CH = BK / 32;                   // fragment chunks per k-tile
    static constexpr int SMEM = 2 * (BM + BN) * 128;

    // acc must be zero-initialised by the caller
    static __device__ __forceinline__ void run(const GemmOperands& p, int m0, int n0, char* smem, f32x4 (&acc)[TM][TN]) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, lr = lane & 15;
        char* sA[2] = {smem, smem + (BM + BN) * 128};
        char* sW[2] = {smem + BM * 128, smem + (BM + BN) * 128 + BM * 128};

        const TA* A = reinterpret_cast<const TA*>(p.A);
        const T* W = reinterpret_cast<const T*>(p.W);

        // per-thread slot assignment: slot s = tid + i*256 -> row = s >> 3, column slot = s & 7
        const int sl = tid & 7;
        const TA* a_src[TM];
        bool a_ok[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (tid >> 3) + i * 32;
            int gm = m0 + row;
            a_ok[i] = gm < p.M;
            if (a_ok[i] && p.a_rows) gm = p.a_rows[gm];
            a_src[i] = A + (size_t)(a_ok[i] ? gm : 0) * p.lda + sl * EPS;
        }
        const T* w_src[TN];
        bool w_ok[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int row = (tid >> 3) + i * 32;
            const int gn = n0 + row;
            w_ok[i] = gn < p.N;
            w_src[i] = W + (size_t)(w_ok[i] ? gn : 0) * p.ldw + sl * EPS;
        }

        RawSlot<T, TA> ra[TM];
        RawSlot<T, T> rw[TN];
        auto gload = [&](int k0) {
            const bool kok = (k0 + sl * EPS) < p.K;     // K is a multiple of EPS (checked on the host)
#pragma unroll
            for (int i = 0; i < TM; ++i) raw_load(ra[i], a_src[i] + k0, a_ok[i] && kok);
#pragma unroll
            for (int i = 0; i < TN; ++i) raw_load(rw[i], w_src[i] + k0, w_ok[i] && kok);
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (tid >> 3) + i * 32;
                *reinterpret_cast<u32x4*>(sA[buf] + row * 128 + ((sl ^ (row & 7)) << 4)) = raw_pack(ra[i]);
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int row = (tid >> 3) + i * 32;
                *reinterpret_cast<u32x4*>(sW[buf] + row * 128 + ((sl ^ (row & 7)) << 4)) = raw_pack(rw[i]);
            }
        };

        const int nt = (p.K + BK - 1) / BK;
        gload(0);
        lstore(0);
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            if (t + 1 < nt) gload((t + 1) * BK);
            const char* a = sA[buf];
            const char* w = sW[buf];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                Frag<T> fa[TM], fw[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) lds_frag(fa[i], a, wm * 16 * TM + i * 16 + lr, c, g);
#pragma unroll
                for (int j = 0; j < TN; ++j) lds_frag(fw[j], w, wn * 16 * TN + j * 16 + lr, c, g);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mma(fw[j], fa[i], acc[i][j]);
            }
            if (t + 1 < nt) lstore(buf ^ 1);
            __syncthreads();
        }
    }
};

}  // namespace pk
