// Third-generation GEMM main loop (round 5): a 256 x 128 macro-tile on ONE 8-wave workgroup per CU whose two 4-wave groups run HALF AN
// ITERATION APART -- "ping-pong" (same operand types, LDS image, fragment reads and accumulator map as gemm_dma.hpp, so gemm_epilogue is shared).
//
// Why (profiles/boundary_probe_r05.txt): the 128 x 128 loop of gemm_dma.hpp moves 64 KB of operands per 4.2 MFLOP and keeps one
// k-tile per workgroup in flight (2 workgroups x 32 KB per CU); at the ~1 us an LDS-DMA piece takes under load that is Little's law for ~35 B/clk
// per CU -- the measured "fill bound".  Every wave of it issues its DMA pieces and then its MFMAs between the same two barriers, so the matrix
// pipe idles through each wait / barrier / issue stretch.  Here
//   * the tile is 256 x 128: 48 KB of operands per 4.2 MFLOP (-25 % fill per flop), wave tiles 64 x 64 (8 fragment reads per 16 MFMAs, was 6 per 8);
//   * the ring holds THREE 48 KB k-tiles (144 KB, one workgroup per CU): up to two of them (96 KB) are in flight while the third is consumed;
//   * waves 0-3 (group 0, rows 0-127 of the tile) and waves 4-7 (group 1, rows 128-255) -- one wave of each group on every SIMD -- alternate:
//         phase A(t):  group 0 runs its 32 MFMAs on k-tile t   |  group 1 issues its 6 LDS-DMA pieces of k-tile t + 2
//         phase B(t):  group 1 runs its 32 MFMAs on k-tile t   |  group 0 issues its 6 pieces of k-tile t + 2
//     with ONE raw s_barrier per phase (gfx950 has one barrier per workgroup: it serves as the rendezvous of both groups), so one group's DMA issue
//     and waiting sit UNDER the other group's matrix work on the same SIMD instead of in front of its own;
//   * the kernel is PERSISTENT: a workgroup walks its tiles, and the first two k-tiles of the NEXT tile are issued before the epilogue of the
//     current one (the ring is dead by then), so a tile's prologue latency rides under the previous tile's stores.
// Every wave always loads the same 6 of a k-tile's 48 pieces of 1 KiB (4 of the 32 A-row pieces, 2 of the 16 W-row pieces); only WHEN it issues
// them depends on its group.  Slot safety: k-tile t + 2 overwrites the slot of k-tile t - 1, whose last reader (group 1, phase B(t - 1)) finished before the barrier
// that opens A(t).  A k-tile is complete before A(t): every wave waits, at the end of B(t - 1), until only its newest 6 pieces are outstanding.
#pragma once
#include "gemm_dma.hpp"

namespace pk {

template <typename T, int TM, int TN, int WM, int WN, int PC = 0>
struct GemmPP {
    static constexpr int ROWB = 128, STAGES = 3;
    static constexpr int NW = WM * WN;
    static_assert(NW == 8 && WM % 2 == 0, "two groups of four waves, split on the rows");
    static constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
    static constexpr int THREADS = 64 * NW;
    static constexpr int PIECES = (BM + BN) / 8, PPW = PIECES / NW;          // 1 KiB pieces per k-tile / per wave
    static_assert(PPW * NW == PIECES, "pieces split evenly over the waves");
    static constexpr int BK = ROWB / (int)sizeof(T);
    static constexpr int CH = BK / 32;
    static constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    static constexpr int SMEM = STAGES * STAGE_BYTES;
    typedef __attribute__((address_space(3))) void* lds_ptr;

    static constexpr int IA = BM / (8 * NW), IW = BN / (8 * NW);              // A / W pieces per wave per k-tile (4 + 2)
    static_assert(IA * 8 * NW == BM && IW * 8 * NW == BN && IA + IW == PPW, "tile rows split evenly over the waves' pieces");

    struct Ctx {                       // per-tile DMA state of one wave: per-lane source offsets (bytes) at k = 0; rows past M / N point out of bounds
        uint32_t offA[IA], offW[IW];
    };

    static __device__ __forceinline__ void setup(const GemmOperands& p, int a_nrows, int m0, int n0, Ctx& c) {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        constexpr int SZ = (int)sizeof(T), SLOTS = ROWB / 16;
        const uint32_t bytesA = (uint32_t)a_nrows * (uint32_t)p.lda * SZ, bytesW = (uint32_t)p.N * (uint32_t)p.ldw * SZ;
        const int lrow = lane / SLOTS, srcslot = (lane % SLOTS) ^ (lrow & 7);
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            int gm = m0 + (wave * IA + i) * 8 + lrow;
            const bool ok = gm < p.M;
            if (ok && p.a_rows) gm = p.a_rows[gm];
            c.offA[i] = ok ? (uint32_t)gm * (uint32_t)p.lda * SZ + srcslot * 16 : bytesA;
        }
#pragma unroll
        for (int i = 0; i < IW; ++i) {
            const int gn = n0 + (wave * IW + i) * 8 + lrow;
            c.offW[i] = gn < p.N ? (uint32_t)gn * (uint32_t)p.ldw * SZ + srcslot * 16 : bytesW;
        }
    }

    // pieces [J0, J1) of this wave's PPW pieces (0 .. IA - 1: A rows, IA .. PPW - 1: W rows) of k-tile kt into ring slot `slot`
    // (j0 / j1 are compile-time constants at every call site: the unrolled loop folds)
    static __device__ __forceinline__ void issue(const GemmOperands& p, int a_nrows, const Ctx& c, int kt, int nt, int slot, char* smem, int j0 = 0, int j1 = PPW) {
        const int lane = threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        constexpr int SZ = (int)sizeof(T), SLOTS = ROWB / 16;
        const uint32_t bytesA = (uint32_t)a_nrows * (uint32_t)p.lda * SZ, bytesW = (uint32_t)p.N * (uint32_t)p.ldw * SZ;
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, bytesA, 0x00020000);
        __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, bytesW, 0x00020000);
        const int srcslot = (lane % SLOTS) ^ ((lane / SLOTS) & 7);
        const int ktail_bytes = (p.K * SZ) % ROWB;
        const bool cut = kt == nt - 1 && ktail_bytes != 0 && srcslot * 16 >= ktail_bytes;     // this lane's piece of the last k-tile lies beyond K
        char* base = smem + slot * STAGE_BYTES;
        const int koff = kt * ROWB;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if (j < j0 || j >= j1) continue;
            if (j < IA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(base + (wave * IA + j) * 1024), 16, cut ? bytesA : c.offA[j], koff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(base + BM * ROWB + (wave * IW + j - IA) * 1024), 16, cut ? bytesW : c.offW[j - IA], koff, 0, 0);
        }
    }

    // first k-tiles of a tile (every wave issues): called at kernel start and, for the next tile, before the current tile's epilogue
    static __device__ __forceinline__ void prologue(const GemmOperands& p, int a_nrows, const Ctx& c, int nt, char* smem) {
        issue(p, a_nrows, c, 0, nt, 0, smem);
        if (nt > 1) issue(p, a_nrows, c, 1, nt, 1, smem);
    }

    // the wave's MFMAs on one k-tile; with PC > 0 the first PC of its DMA pieces of k-tile `kt2` are issued BETWEEN the rows of MFMAs (a piece costs
    // ~60 cycles of issue among MFMAs the matrix pipe is still executing, 100-185 in a phase of its own: MI355X_MICROARCH.md), evenly spread
    static __device__ __forceinline__ void compute(const GemmOperands& p, int a_nrows, const Ctx& cx, int kt2, int nt, int nslot, bool more, char* smem,
                                                   const char* a, int wm, int wn, int lr, int g, f32x4 (&acc)[TM][TN]) {
        const char* w = a + BM * ROWB;
        constexpr int POS = CH * TM;                                            // rows of MFMAs per k-tile
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            Frag<T> fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) lds_frag_a<T>(fa[i], a, wm * 16 * TM + i * 16 + lr, c, g);
#pragma unroll
            for (int j = 0; j < TN; ++j) lds_frag_w<T>(fw[j], w, wn * 16 * TN + j * 16 + lr, c, g);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mma(fw[j], fa[i], acc[i][j]);
                if constexpr (PC > 0) {
#pragma unroll
                    for (int q = 0; q < PC; ++q)
                        if (((2 * q + 1) * POS) / (2 * PC) == c * TM + i && more) issue_one(q, p, a_nrows, cx, kt2, nt, nslot, smem);
                }
            }
        }
    }

    static __device__ __forceinline__ void issue_one(int q, const GemmOperands& p, int a_nrows, const Ctx& c, int kt, int nt, int slot, char* smem) {
        issue(p, a_nrows, c, kt, nt, slot, smem, q, q + 1);
    }

    // main loop of one tile whose prologue() has been issued.  acc zero-initialised by the caller.  Ends with the ring dead (barrier).
    static __device__ __forceinline__ void run(const GemmOperands& p, int a_nrows, const Ctx& c, int nt, char* smem, f32x4 (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63, g = lane >> 4, lr = lane & 15;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int wm = wave / WN, wn = wave % WN;
        const bool grp1 = wave >= NW / 2;                                       // (wave-uniform)
        if (nt > 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();                    // k-tile 0 landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                                           // ... and everybody's
        // Two straight-line loops, one per group, chosen once (wave-uniform): inside a loop the accumulators are updated in place.  (One loop with
        // `if (group) compute else issue` in both phases made hipcc keep two copies of the 64 accumulator registers and move one onto the other
        // after every phase: 256 VGPRs + spills.)  Both loops execute the same two barriers per k-tile.
        int slot = 0;
        if (!grp1) {
#pragma unroll 1
            for (int kt = 0; kt < nt; ++kt) {
                const char* a = smem + slot * STAGE_BYTES;
                const bool more = kt + 2 < nt;
                const int nslot = slot == 0 ? 2 : slot - 1;                     // (kt + 2) % 3
                slot = slot == 2 ? 0 : slot + 1;
                compute(p, a_nrows, c, kt + 2, nt, nslot, more, smem, a, wm, wn, lr, g, acc);      // phase A: this group computes (+ its first PC pieces), group 1 feeds the ring
                __builtin_amdgcn_s_barrier();
                if (more) issue(p, a_nrows, c, kt + 2, nt, nslot, smem, PC, PPW);                  // phase B: group 1 computes, this group feeds the ring
                // k-tile kt + 1 must be complete before phase A(kt + 1): only this wave's pieces of kt + 2 may stay in flight
                if (more) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
        } else {
#pragma unroll 1
            for (int kt = 0; kt < nt; ++kt) {
                const char* a = smem + slot * STAGE_BYTES;
                const bool more = kt + 2 < nt;
                const int nslot = slot == 0 ? 2 : slot - 1;
                slot = slot == 2 ? 0 : slot + 1;
                if (more) issue(p, a_nrows, c, kt + 2, nt, nslot, smem, PC, PPW);                  // phase A
                __builtin_amdgcn_s_barrier();
                compute(p, a_nrows, c, kt + 2, nt, nslot, more, smem, a, wm, wn, lr, g, acc);      // phase B
                if (more) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
        }
    }
};

}  // namespace pk
