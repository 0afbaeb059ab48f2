// DMA-fed MFMA GEMM main loop (second-generation core; same tile/fragment geometry and LDS image as gemm_core.hpp):
//   acc[m][n] = sum_k A[m][k] * W[n][k],  A and W both of type T, W zero-padded along K to a multiple of BK.
//
// Operand tiles go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, no VGPR round
// trip) into a ring of STAGES k-tiles; STAGES-1 tiles are in flight ahead of the MFMAs, retired with counted
// s_waitcnt vmcnt(N) + ONE raw s_barrier per k-tile (never __syncthreads(): it would drain the ring).
// An LDS-DMA writes wave-uniform base + lane*16, so the 16-B-slot XOR swizzle that makes ds_read_b128 conflict-free
// is applied on the per-lane SOURCE address; the read side XORs the same way.
//   ROWB = 128 (k-tile of 64 bf16 / 32 f32): slot ^ (row & 7)
// Rows past M / N are fetched through the buffer descriptor's bounds check (their per-lane offset is the descriptor size:
// they read as 0).  The K tail (K not a multiple of the k-tile) is cut the same way: in the LAST k-tile every 16-byte piece
// that starts at or beyond K is pointed out of bounds, so A never reads into its next row (a NaN / Inf there would survive the
// multiplication with W's zero padding) or past the end of its allocation.  The k-tile advance goes through the scalar
// offset, which the bounds check does not include -- hence the explicit per-lane redirect instead of relying on it.
#pragma once
#include "gemm_core.hpp"

namespace pk {

// ---- row statistics of the A operand, taken from the MFMA fragments as they pass (LayerNorm folded into the GEMM, see
// gemm.hip): a lane holds 8 k-elements of ONE row per fragment chunk, so sum / sum of squares are a few VALU ops beside the
// MFMAs (bf16: v_dot2c_f32_bf16 against (1, 1) and against itself) and a 4-lane-group shuffle at the end.
// v_dot2c_f32_bf16 d, a, b : d += a.lo * b.lo + a.hi * b.hi (f32 accumulate).  Issued through inline asm on the raw 32-bit words:
// with __builtin_amdgcn_fdot2_f32_bf16 on elements of the u32x4 fragment hipcc (ROCm 7.2) selected the FIRST word four times
// (every v_dot2c of a fragment read the same VGPR; found by the statistics coming out wrong, confirmed in the ISA).
__device__ __forceinline__ void dot2acc_bf16(float& d, uint32_t a, uint32_t b) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void frag_stats(const Frag<bf16>& f, float& s, float& q, bool want_sq) {
    const uint32_t w0 = f.v[0], w1 = f.v[1], w2 = f.v[2], w3 = f.v[3];
    const uint32_t ones = 0x3f803f80u;                       // (1.0, 1.0) in bf16
    dot2acc_bf16(s, w0, ones); dot2acc_bf16(s, w1, ones); dot2acc_bf16(s, w2, ones); dot2acc_bf16(s, w3, ones);
    if (want_sq) { dot2acc_bf16(q, w0, w0); dot2acc_bf16(q, w1, w1); dot2acc_bf16(q, w2, w2); dot2acc_bf16(q, w3, w3); }
}
__device__ __forceinline__ void frag_stats(const Frag<float>& f, float& s, float& q, bool want_sq) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s += f.lo[e] + f.hi[e];
        if (want_sq) q = fmaf(f.lo[e], f.lo[e], fmaf(f.hi[e], f.hi[e], q));
    }
}

#ifdef PK_TIMELINE
// instrumented build only (tools/build_alt.sh tl -DPK_TIMELINE; tools/gemm_timeline.py): s_memtime stamps of wave 0 of a few workgroups
__device__ unsigned long long pk_tl[8 * 5 * 40];
#define PK_TL(slot) do { if (tl_on && kt < 38) pk_tl[(tl_wg * 40 + kt) * 5 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ __forceinline__ int pk_tl_wg() {
    return blockIdx.x == 0 ? 0 : blockIdx.x == 8 ? 1 : blockIdx.x == 1 ? 2 : blockIdx.x == gridDim.x / 2 ? 3 :
           blockIdx.x == gridDim.x - 8 ? 4 : blockIdx.x == 256 ? 5 : blockIdx.x == 512 ? 6 : blockIdx.x == 1024 ? 7 : -1;
}
// row 39 of a workgroup's table: [0] kernel entry, [1] main loop done, [2] epilogue done
#define PK_TL_KERNEL(slot) do { const int w_ = pk_tl_wg(); if (w_ >= 0 && threadIdx.x == 0) pk_tl[(w_ * 40 + 39) * 5 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PK_TL_KERNEL(slot) do {} while (0)
#define PK_TL(slot) do {} while (0)
#endif

__device__ __forceinline__ void frag_stats(const Frag<bf16x3>& f, float& s, float& q, bool want_sq) {      // hi + lo planes: the f32 row sums to ~2^-17
    const uint32_t ones = 0x3f803f80u;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        dot2acc_bf16(s, f.hi[w], ones); dot2acc_bf16(s, f.lo[w], ones);
        if (want_sq) { dot2acc_bf16(q, f.hi[w], f.hi[w]); dot2acc_bf16(q, f.hi[w], f.lo[w]); dot2acc_bf16(q, f.hi[w], f.lo[w]); }
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TM x TN MFMA tiles per wave, WM x WN compute waves per workgroup, ROWB bytes of k per LDS row.
// PW = 0: every wave both feeds the ring and computes (64 * WM * WN threads).
// PW > 0: role split -- PW extra PRODUCER waves (wave ids >= WM*WN) issue all the LDS-DMA pieces and wait for them, the
//         WM*WN CONSUMER waves only do ds_read + MFMA (64 * (WM*WN + PW) threads).  A DMA issue stalls its wave for
//         ~60-180 cycles while the texture path is busy; with the roles split that stall no longer sits in front of
//         the consumer's MFMAs on the same SIMD (MI355X: separate waves issue independently).
template <typename T, int TM, int TN, int WM, int WN, int STAGES, int ROWB = 128, int PW = 0>
struct GemmDma {
    static_assert(ROWB == 128, "k-tile = 128 bytes per row (the 64-byte k-tile variants lost the round-1 sweep and were removed)");
    static constexpr int NW = WM * WN;
    static constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
    static constexpr int RPI = 1024 / ROWB;                             // rows per DMA wave-instruction
    static constexpr int SLOTS = ROWB / 16;
    static constexpr int NLW = PW > 0 ? PW : NW;                        // waves that issue DMA
    static constexpr int THREADS = 64 * (NW + PW);
    static constexpr int IA = BM / (RPI * NLW), IW = BN / (RPI * NLW);  // DMA instructions per loading wave per k-tile
    static_assert(IA * RPI * NLW == BM && IW * RPI * NLW == BN, "tile rows must split evenly over the loading waves' DMA pieces");
    static constexpr int BK = ROWB / (int)sizeof(T);
    static constexpr int CH = BK / 32;
    static constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    static constexpr int SMEM = STAGES * STAGE_BYTES;
    static constexpr int IPW = IA + IW;
    static_assert((STAGES - 2) * IPW <= 63, "vmcnt is a 6-bit counter");

    typedef __attribute__((address_space(3))) void* lds_ptr;

    template <int TILES> static __device__ __forceinline__ void wait_tiles() { wait_vmcnt<TILES * IPW>(); }

    static __device__ __forceinline__ void wait_outstanding(int tiles) {
        // `tiles` k-tiles issued after the one about to be consumed may stay in flight
        switch (tiles) {
            case 0: wait_tiles<0>(); break;
            case 1: wait_tiles<1>(); break;
            case 2: wait_tiles<(STAGES > 3 ? 2 : 0)>(); break;
            case 3: wait_tiles<(STAGES > 4 ? 3 : 0)>(); break;
            case 4: wait_tiles<(STAGES > 5 ? 4 : 0)>(); break;
            case 5: wait_tiles<(STAGES > 6 ? 5 : 0)>(); break;
            default: wait_tiles<(STAGES > 7 ? 6 : 0)>(); break;
        }
    }

    // acc must be zero-initialised by the caller. a_nrows = number of physical rows behind p.A (for the bounds check).
    // Returns true for the waves that hold accumulators (all of them when PW == 0); producer waves must skip the epilogue.
    static __device__ __forceinline__ bool run(const GemmOperands& p, int a_nrows, int m0, int n0, char* smem, f32x4 (&acc)[TM][TN]) {
        float dummy[TM];
        return run_stats<0>(p, a_nrows, m0, n0, smem, acc, dummy, dummy);
    }

    // STATS = 0: plain; 1: rsum[i] = sum_k A[row][k] of the lane's row (wm*16*TM + i*16 + lr); 2: also rsq[i] = sum_k A[row][k]^2.
    // Every compute wave gets the statistics of its own rows (waves of one row group compute them redundantly).
    template <int STATS>
    static __device__ __forceinline__ bool run_stats(const GemmOperands& p, int a_nrows, int m0, int n0, char* smem, f32x4 (&acc)[TM][TN],
                                                     float (&rsum)[TM], float (&rsq)[TM]) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool loads = PW == 0 || wave_all >= NW;
        const bool computes = PW == 0 || wave_all < NW;
        const int wave = PW == 0 ? wave_all : (loads ? wave_all - NW : 0);      // index among the loading waves
        const int cw = computes ? wave_all : 0;                                 // index among the compute waves
        const int wm = cw / WN, wn = cw % WN, g = lane >> 4, lr = lane & 15;
        constexpr int SZ = (int)sizeof(T);

        const uint32_t bytesA = (uint32_t)a_nrows * (uint32_t)p.lda * SZ;
        const uint32_t bytesW = (uint32_t)p.N * (uint32_t)p.ldw * SZ;
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, bytesA, 0x00020000);
        __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, bytesW, 0x00020000);

        // per-lane source offsets (bytes) at k = 0; the k-tile advance goes through the scalar offset.
        // lane -> (row within the instruction's RPI rows, 16-byte slot); the slot is swizzled on the SOURCE side
        const int lrow = lane / SLOTS, lslot = lane % SLOTS;
        const int srcslot = lslot ^ (lrow & 7);
        uint32_t offA[IA], offW[IW];
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int row = (wave * IA + i) * RPI + lrow;
            int gm = m0 + row;
            const bool ok = gm < p.M;
            if (ok && p.a_rows) gm = p.a_rows[gm];
            offA[i] = ok ? (uint32_t)gm * (uint32_t)p.lda * SZ + srcslot * 16 : bytesA;
        }
#pragma unroll
        for (int i = 0; i < IW; ++i) {
            const int row = (wave * IW + i) * RPI + lrow;
            const int gn = n0 + row + ((p.w_gap_from > 0 && row >= p.w_gap_from) ? p.w_gap_rows : 0);
            offW[i] = gn < p.N ? (uint32_t)gn * (uint32_t)p.ldw * SZ + srcslot * 16 : bytesW;
        }

        // k-rotation: every workgroup of a launch reads the SAME 128-byte column slab of A and W at the same time, and
        // with power-of-two row strides (1 KiB for K = 512 bf16) a slab lives on a fraction of the L2 channels: the
        // workgroups queue on those while the others idle.  Starting workgroup b at k-tile (b >> 3) % nt (b >> 3 = its
        // index inside its XCD) spreads the concurrent slabs over all channels; the sum over k only changes order.
        const int nt = (p.K + BK - 1) / BK;
        const int rot = (p.krot && nt > 1) ? (int)(((blockIdx.x >> 3) + blockIdx.y) % (unsigned)nt) : 0;
        const int ktail_bytes = (p.K * SZ) % ROWB;                    // bytes of the last k-tile that exist (0: K is a multiple of BK)
        const bool slot_in_tail = ktail_bytes == 0 || srcslot * 16 < ktail_bytes;      // K % (16 / SZ) == 0 (host check): whole pieces
        auto issue = [&](int kt, int slot) {
            char* base = smem + slot * STAGE_BYTES;
            int kk = kt + rot;
            if (kk >= nt) kk -= nt;
            const int koff = kk * ROWB;
            const bool cut = kk == nt - 1 && !slot_in_tail;          // this lane's piece of the last k-tile lies beyond K
#pragma unroll
            for (int i = 0; i < IA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(base + (wave * IA + i) * 1024), 16, cut ? bytesA : offA[i], koff, 0, 0);
#pragma unroll
            for (int i = 0; i < IW; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(base + BM * ROWB + (wave * IW + i) * 1024), 16, offW[i], koff, 0, 0);
        };

        if (STATS) {
#pragma unroll
            for (int i = 0; i < TM; ++i) { rsum[i] = 0.f; rsq[i] = 0.f; }
        }
        const int pre = nt < STAGES - 1 ? nt : STAGES - 1;
        if (loads)
            for (int s = 0; s < pre; ++s) issue(s, s);
        if (PW > 0 && computes) __builtin_amdgcn_s_setprio(1);      // consumers win issue arbitration against their SIMD's producer
#ifdef PK_TIMELINE
        const int tl_wg = pk_tl_wg();
        const bool tl_on = tl_wg >= 0 && tid == 0;
#endif
        for (int kt = 0; kt < nt; ++kt) {
            const int issued = (kt + STAGES - 1 < nt) ? kt + STAGES - 1 : nt;      // tiles issued so far
            PK_TL(0);
            if (loads) wait_outstanding(issued - (kt + 1));
            PK_TL(1);
            __builtin_amdgcn_s_barrier();                 // tile kt landed for every wave; everyone is done with tile kt-1
            PK_TL(2);
            if (loads && kt + STAGES - 1 < nt) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
            PK_TL(3);
            if (!computes) continue;
            const char* a = smem + (kt % STAGES) * STAGE_BYTES;
            const char* w = a + BM * ROWB;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                Frag<T> fa[TM], fw[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) lds_frag_a<T>(fa[i], a, wm * 16 * TM + i * 16 + lr, c, g);
#pragma unroll
                for (int j = 0; j < TN; ++j) lds_frag_w<T>(fw[j], w, wn * 16 * TN + j * 16 + lr, c, g);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mma(fw[j], fa[i], acc[i][j]);
                if (STATS) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) frag_stats(fa[i], rsum[i], rsq[i], STATS > 1);
                }
            }
            PK_TL(4);
        }
        if (STATS) {                                      // fold the 4 lane groups (k = g*8 + 0..7 of every chunk) of each row
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                rsum[i] += __shfl_xor(rsum[i], 16, 64);
                rsum[i] += __shfl_xor(rsum[i], 32, 64);
                if (STATS > 1) {
                    rsq[i] += __shfl_xor(rsq[i], 16, 64);
                    rsq[i] += __shfl_xor(rsq[i], 32, 64);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // the ring is dead: callers may reuse smem
        return computes;
    }
};

}  // namespace pk
