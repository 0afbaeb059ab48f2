// Third-generation main loop: 256 x 256 macro-tile, 8 waves (2 x 4), ONE workgroup per CU, 16 MFMAs per phase, 4 phases per
// k-tile, LDS-DMA HALF-tiles (128 rows x 128 B = 16 KB) in a ring of 8, counted vmcnt ONCE per k-tile, two 4-wave groups a
// barrier apart.  Same operand image (128-byte LDS rows, 16-B slot ^ (row & 7) on the DMA source and on the read), same fragment
// geometry and the same transposed product (lane = 4 consecutive output columns of one row) as gemm_dma.hpp, so every epilogue of
// gemm.hip applies unchanged to a quadrant of the tile.
//
// Why this shape (profiles/gemm_vendor_r06.txt): the 128 x 128 ring waits for a WHOLE k-tile, passes a workgroup
// barrier and consumes it; its MFMA pipe idles through wait + barrier + DMA issue (28 % busy, profiles/gemm_pmc_r02.txt) and two
// co-resident workgroups only partly hide that in each other.  Here
//   * a wave owns 128 x 64 of the output as 2 x 2 QUADRANTS of 64 x 32: quadrant (a, b) = rows a*128 + wr*64 .. +64, columns
//     b*128 + wc*32 .. +32 -- every wave reads from all four half-tiles (A0, A1, B0, B1) of a k-tile, 12 KB of fragments per 64 MFMAs
//     (the 128 x 128 8-wave loop: 6 KB per 16);
//   * phase j of a k-tile = [load part: this phase's fragment reads + the 2 DMA pieces of ONE half-tile 7 half-tiles ahead] s_barrier
//     [16 MFMAs of one quadrant] s_barrier.  Waves 4-7 run one barrier behind waves 0-3, and wave w shares its SIMD with wave w + 4,
//     so on every SIMD one wave's MFMAs run beside the other's LDS / DMA issue (matrix beside memory, the complementary pairing);
//   * half-tile h of the stream (4 per k-tile in the order B0, A0, B1, A1) lives in ring slot h & 7 and is re-filled at phase h + 1:
//     B0 is read in phase 0 only (its fragments stay in registers for phase 3), A0 in phase 0, B1 in phase 1, A1 in phase 2, so every
//     slot is dead a full phase before its DMA is even issued -- except B0's, whose 4 reads are retired by lgkmcnt(8) BEFORE phase 0's
//     first barrier (the later group's reads then precede the earlier group's re-fill issue by a barrier);
//   * the only vmcnt of the loop sits in phase 3's load part: it leaves the 3 newest half-tiles (6 pieces of k-tile t + 2) in flight and
//     retires k-tile t + 1, whose first read is a full barrier later for both groups.
// Accumulators: acc[a][b][i][j] = 16 x 16 block (i, j) of quadrant (a, b); C row = m0 + a*128 + wr*64 + i*16 + (lane & 15),
// column = n0 + b*128 + wc*32 + j*16 + (lane >> 4)*4 + r -- i.e. gemm_epilogue<T, 4, 2, 4> at (m0 + a*128, n0 + b*128).
#pragma once
#include <type_traits>
#include "gemm_dma.hpp"

namespace pk {

// FLAGS: ablation switches of tools/gemm_bench.py (0 in the product): 1 no in-loop DMA (stale operands), 2 no in-loop fragment reads, 4 no group stagger,
// 8 no s_setprio, 16 fragment reads retired BEFORE the phase's first barrier
template <typename T, int FLAGS = 0>
struct GemmP8 {
    // T = bf16: both operand tiles hold bf16 (k-tile 64).  T = bf16x3 (split-bf16, common.hpp): the A tile holds f32 rows (k-tile 32), split into (hi, lo) planes in
    // registers right after the ds_reads -- in the LOAD part of a phase, beside the partner's MFMAs -- and the W tile holds the host-packed planes [hi x 32 | lo x 32];
    // a fragment pair is three MFMAs, so a phase carries 48 of them against the same 64 KB of DMA per k-tile: the fill bytes per MFMA cycle are 2/3 of bf16's.
    static constexpr bool SPLIT = sizeof(T) == 4;
    static constexpr int BM = 256, BN = 256, THREADS = 512, ROWB = 128;
    static constexpr int BK = ROWB / (int)sizeof(T);
    static constexpr int CH = BK / 32;
    static constexpr int HALF = 128 * ROWB;               // one half-tile: 16 KB = 16 DMA pieces of 1 KiB, 2 per wave
    static constexpr int SMEM = 8 * HALF;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef f32x4 QuadAcc16[4][2];
    typedef f32x16 QuadAcc32[2];
    typedef typename std::conditional<(FLAGS & 32) != 0, QuadAcc32, QuadAcc16>::type QuadAcc;
    typedef QuadAcc Acc[2][2];

    struct Ctx {
        __amdgpu_buffer_rsrc_t rsA, rsW;
        uint32_t bytesA, bytesW;
        uint32_t offA[2][2], offW[2][2];                 // [half][piece] per-lane source offsets at k = 0
        int nt, rot;
        bool slot_in_tail;
        char* smem;
        int piece0;                                       // wave * 2: first DMA piece of a half-tile this wave issues
        int rdA, rdB;                                     // per-lane LDS byte offsets of the fragment reads inside a half-tile, chunk 0 (chunk 1: ^ 64)
        u32x4 rawA;                                       // (timing experiments) the A descriptor as plain dwords for inline-asm buffer loads
        mutable f32x4 junk[4];
    };

    // FLAGS & 4096 (timing only): what an ORDINARY 16-byte-per-lane buffer load to VGPRs costs a wave that is issuing MFMAs (the staging path of a one-wave-per-
    // SIMD design).  Inline asm: the compiler sees no memory operation and inserts no s_waitcnt; the destination is kept alive and never read.
    static __device__ __forceinline__ void vgpr_load(const Ctx& c, f32x4& dst, uint32_t voff, int koff) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(c.rawA), "s"(koff) : "memory");
    }

    // half-tile TYPE (0: B0, 1: A0, 2: B1, 3: A1) of k-tile kt into ring slot SLOT
    static __device__ __forceinline__ void issue(const int TYPE, const int SLOT, const Ctx& c, int kt, bool inloop = true, const int pieces = 3) {
        if ((FLAGS & 1) && inloop) return;
        int kk = kt + c.rot;
        if (kk >= c.nt) kk -= c.nt;
        const int koff = kk * ROWB;
        const bool cut = kk == c.nt - 1 && !c.slot_in_tail;
        char* base = c.smem + SLOT * HALF + c.piece0 * 1024;
        const int h = TYPE >> 1;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            if (!((pieces >> pi) & 1)) continue;
            if (TYPE & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsA, (lds_ptr)(base + pi * 1024), 16, cut ? c.bytesA : c.offA[h][pi], koff, 0, 0);
            else          __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsW, (lds_ptr)(base + pi * 1024), 16, c.offW[h][pi], koff, 0, 0);
        }
    }

    // NF = 4: A fragments (rows of the wave's 64-row sub-tile), NF = 2: W fragments (its 32 columns)
    template <int SLOT, int NF>
    static __device__ __forceinline__ void read_frags(const Ctx& c, int rd, Frag<T> (&f)[NF][CH], bool inloop = true) {
        if ((FLAGS & 2) && inloop) return;
        const char* s = c.smem + SLOT * HALF;
        if constexpr (!SPLIT) {
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) f[i][ch].v = *reinterpret_cast<const u32x4*>(s + ((rd + i * 16 * ROWB) ^ (ch * 64)));
        } else if constexpr (NF == 2) {
            // W planes: k = g*8 .. +7 of a row is slot g (hi) and slot 4 + g (lo), both ^ (row & 7): the lo slot is the hi slot ^ 4, i.e. byte ^ 64
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                f[i][0].hi = *reinterpret_cast<const u32x4*>(s + (rd + i * 16 * ROWB));
                f[i][0].lo = *reinterpret_cast<const u32x4*>(s + ((rd + i * 16 * ROWB) ^ 64));
            }
        } else {
            // f32 A rows: k = g*8 .. +7 is slots 2g, 2g + 1 (^ (row & 7)): the second is the first ^ 1, i.e. byte ^ 16.  All reads first, then the splits.
            f32x4 lo4[NF], hi4[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                lo4[i] = *reinterpret_cast<const f32x4*>(s + (rd + i * 16 * ROWB));
                hi4[i] = *reinterpret_cast<const f32x4*>(s + ((rd + i * 16 * ROWB) ^ 16));
            }
#pragma unroll
            for (int i = 0; i < NF; ++i) split8(lo4[i], hi4[i], f[i][0].hi, f[i][0].lo);
        }
    }

    static __device__ __forceinline__ void quad(f32x4 (&q)[4][2], const Frag<T> (&fa)[4][CH], const Frag<T> (&fb)[2][CH]) {
        if (!(FLAGS & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ch = 0; ch < CH; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) q[i][j] = mma(fb[j][ch], fa[i][ch], q[i][j]);
        if (!(FLAGS & 8)) __builtin_amdgcn_s_setprio(0);
    }

    // FLAGS & 32 (timing experiment only, results are NOT a GEMM): the same operand registers fed to v_mfma_f32_32x32x16_bf16, 8 per quadrant
    static __device__ __forceinline__ void quad(f32x16 (&q)[2], const Frag<T> (&fa)[4][CH], const Frag<T> (&fb)[2][CH]) {
        if (!(FLAGS & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            q[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[ks >> 1][ks & 1].v), __builtin_bit_cast(bf16x8_t, fa[ks >> 1][ks & 1].v), q[0], 0, 0, 0);
            q[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[ks >> 1][ks & 1].v), __builtin_bit_cast(bf16x8_t, fa[2 + (ks >> 1)][ks & 1].v), q[1], 0, 0, 0);
        }
        if (!(FLAGS & 8)) __builtin_amdgcn_s_setprio(0);
    }

    static __device__ __forceinline__ void bar() {
        __builtin_amdgcn_sched_barrier(0);
        if (!(FLAGS & 256)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // one k-tile (ring parity E = kt & 1)
    template <int E>
    static __device__ __forceinline__ void ktile(const Ctx& c, int kt, Acc& acc, Frag<T> (&fa)[4][CH], Frag<T> (&fb0)[2][CH], Frag<T> (&fb1)[2][CH]) {
        constexpr int S = 4 * E, SN = 4 * (E ^ 1);
        const bool more1 = kt + 1 < c.nt, more2 = kt + 2 < c.nt;
        // ---- phase 0: B0 + A0 fragments; DMA: A1 of k-tile kt + 1
        read_frags<S + 0, 2>(c, c.rdB, fb0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags<S + 1, 4>(c, c.rdA, fa);
        if (more1) issue(3, SN + 3, c, kt + 1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 * CH) : "memory");      // the B0 reads are back: slot S + 0 may be re-filled from phase 1 on
        if (FLAGS & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quad(acc[0][0], fa, fb0);
        bar();
        // ---- phase 1: B1 fragments; DMA: B0 of k-tile kt + 2
        read_frags<S + 2, 2>(c, c.rdB, fb1);
        if (more2) issue(0, S + 0, c, kt + 2);
        if (FLAGS & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quad(acc[0][1], fa, fb1);
        bar();
        // ---- phase 2: A1 fragments; DMA: A0 of k-tile kt + 2
        read_frags<S + 3, 4>(c, c.rdA, fa);
        if (more2) issue(1, S + 1, c, kt + 2);
        if (FLAGS & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        quad(acc[1][1], fa, fb1);
        bar();
        // ---- phase 3: no reads (B0 is still in registers); DMA: B1 of k-tile kt + 2; retire k-tile kt + 1
        if (FLAGS & 1) {}
        else if (more2) { issue(2, S + 2, c, kt + 2); wait_vmcnt<6>(); }
        else wait_vmcnt<0>();
        bar();
        quad(acc[1][0], fa, fb0);
        bar();
    }

    // acc must be zero-initialised by the caller.  a_nrows = physical rows behind p.A.
    static __device__ __forceinline__ void setup(Ctx& c, const GemmOperands& p, int a_nrows, int m0, int n0, char* smem) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wr = wave >> 2, wc = wave & 3, g = lane >> 4, lr = lane & 15;
        constexpr int SZ = (int)sizeof(T);
        c.smem = smem;
        c.piece0 = wave * 2;
        c.bytesA = (uint32_t)a_nrows * (uint32_t)p.lda * SZ;
        c.bytesW = (uint32_t)p.N * (uint32_t)p.ldw * SZ;
        c.rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, c.bytesA, 0x00020000);
        {
            const uint64_t ab = reinterpret_cast<uint64_t>(p.A);
            c.rawA = u32x4{(uint32_t)ab, (uint32_t)(ab >> 32) & 0xFFFFu, c.bytesA, 0x00020000u};
        }
        c.rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, c.bytesW, 0x00020000);
        const int lrow = lane >> 3, lslot = lane & 7, srcslot = lslot ^ (lrow & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                const int row = h * 128 + wave * 16 + pi * 8 + lrow;
                int gm = m0 + row;
                const bool ok = gm < p.M;
                if (ok && p.a_rows) gm = p.a_rows[gm];
                c.offA[h][pi] = ok ? (uint32_t)gm * (uint32_t)p.lda * SZ + srcslot * 16 : c.bytesA;
                const int gn = n0 + row;
                c.offW[h][pi] = gn < p.N ? (uint32_t)gn * (uint32_t)p.ldw * SZ + srcslot * 16 : c.bytesW;
            }
        c.nt = (p.K + BK - 1) / BK;
        c.rot = (p.krot && c.nt > 1) ? (int)(((blockIdx.x >> 3) + blockIdx.y) % (unsigned)c.nt) : 0;
        const int ktail_bytes = (p.K * SZ) % ROWB;
        c.slot_in_tail = ktail_bytes == 0 || srcslot * 16 < ktail_bytes;
        // fragment reads: row (base + lr) of a half-tile, 16-B slot (ch*4 + g) ^ (row & 7); row bases are multiples of 16, so row & 7 = lr & 7
        c.rdA = (wr * 64 + lr) * ROWB + (((SPLIT ? 2 * g : g) ^ (lr & 7)) << 4);
        c.rdB = (wc * 32 + lr) * ROWB + ((g ^ (lr & 7)) << 4);
    }

    static __device__ __forceinline__ void run(const GemmOperands& p, int a_nrows, int m0, int n0, char* smem, Acc& acc) {
        Ctx c;
        setup(c, p, a_nrows, m0, n0, smem);
        const int wr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
        // ---- prologue: k-tile 0 whole, B0 / A0 / B1 of k-tile 1
        issue(0, 0, c, 0, false); issue(1, 1, c, 0, false); issue(2, 2, c, 0, false); issue(3, 3, c, 0, false);
        if (c.nt > 1) { issue(0, 4, c, 1, false); issue(1, 5, c, 1, false); issue(2, 6, c, 1, false); wait_vmcnt<6>(); }
        else wait_vmcnt<0>();
        bar();
        if (wr == 1 && !(FLAGS & 4)) bar();               // waves 4-7 run one barrier behind

        Frag<T> fa[4][CH], fb0[2][CH], fb1[2][CH];
        if (FLAGS & 2) { read_frags<0, 2>(c, c.rdB, fb0, false); read_frags<2, 2>(c, c.rdB, fb1, false); read_frags<1, 4>(c, c.rdA, fa, false); }
        for (int kt = 0; kt < c.nt; kt += 2) {
            ktile<0>(c, kt, acc, fa, fb0, fb1);
            if (kt + 1 < c.nt) ktile<1>(c, kt + 1, acc, fa, fb0, fb1);
        }
        if (wr == 0 && !(FLAGS & 4)) bar();
        bar();                                            // the ring is dead: callers may reuse smem
    }
};

// ---- the same tile with TWO phases per k-tile (32 MFMAs each, 4 barriers per k-tile instead of 8) ------------------------------------------
// Measured on the 8-phase loop (profiles/gemm_p8_r06.txt): with NO loads at all its barrier skeleton tops out at 1.6 PF -- every s_barrier costs the
// SIMD a ~60-80-cycle MFMA bubble, 8 of them per 2048 MFMA-cycles -- so the phases are made twice as long:
//   phase 0: reads B0, B1, A0 (16 x ds_read_b128), issues A1 of k-tile t + 1, 32 MFMAs of quadrants (0,0), (0,1);
//   phase 1: reads A1 (8), issues B0, B1, A0 of k-tile t + 2 (their slots died with phase 0's reads), 32 MFMAs of (1,1), (1,0).
// Every fragment read is retired (lgkmcnt(0)) BEFORE its phase's first barrier, so the later group's reads of a slot precede the earlier group's re-fill
// of it by a barrier.  Ring slots of k-tile t: 4 * (t & 1) + {0: B0, 1: B1, 2: A0, 3: A1}.  One vmcnt(8) per phase: phase 0 retires A1 of THIS k-tile
// (issued a whole k-tile earlier, read in phase 1), phase 1 retires B0 / B1 / A0 of k-tile t + 1 (issued a k-tile earlier, read in the next phase 0).
template <typename T, int FLAGS = 0>
struct GemmP4 : GemmP8<T, FLAGS> {
    using P = GemmP8<T, FLAGS>;
    using typename P::Ctx;
    using typename P::Acc;
    static constexpr int CH = P::CH;

    template <int E>
    static __device__ __forceinline__ void ktile(const Ctx& c, int kt, Acc& acc, Frag<T> (&fa)[4][CH], Frag<T> (&fb0)[2][CH], Frag<T> (&fb1)[2][CH]) {
        constexpr int S = 4 * E, SN = 4 * (E ^ 1);
        const bool more1 = kt + 1 < c.nt, more2 = kt + 2 < c.nt;
        if constexpr ((FLAGS & 2048) != 0) {
            // DMA issue spread: none beside phase 0's 16 reads; A1 of k-tile t + 1 between phase 0's MFMAs; B0 / B1 of k-tile t + 2 beside phase 1's reads;
            // A0 of k-tile t + 2 between phase 1's MFMAs
            P::template read_frags<S + 0, 2>(c, c.rdB, fb0);
            P::template read_frags<S + 1, 2>(c, c.rdB, fb1);
            P::template read_frags<S + 2, 4>(c, c.rdA, fa);
            if (!(FLAGS & 65)) { if (more1) wait_vmcnt<6>(); else wait_vmcnt<0>(); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            P::bar();
            P::quad(acc[0][0], fa, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (more1) P::issue(3, SN + 3, c, kt + 1, true, 1);
            __builtin_amdgcn_sched_barrier(0);
            P::quad(acc[0][1], fa, fb1);
            __builtin_amdgcn_sched_barrier(0);
            if (more1) P::issue(3, SN + 3, c, kt + 1, true, 2);
            P::bar();
            if (more2) { P::issue(0, S + 0, c, kt + 2); P::issue(2, S + 1, c, kt + 2); }
            P::template read_frags<S + 3, 4>(c, c.rdA, fa);
            if (!(FLAGS & 65)) { if (more2) wait_vmcnt<6>(); else if (more1) wait_vmcnt<2>(); else wait_vmcnt<0>(); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            P::bar();
            P::quad(acc[1][1], fa, fb1);
            __builtin_amdgcn_sched_barrier(0);
            if (more2) P::issue(1, S + 2, c, kt + 2, true, 1);
            __builtin_amdgcn_sched_barrier(0);
            P::quad(acc[1][0], fa, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (more2) P::issue(1, S + 2, c, kt + 2, true, 2);
            P::bar();
            return;
        }
        // ---- phase 0   (FLAGS & 1024: the phase's reads are issued BEFORE its DMA pieces; & 512: A0 of k-tile t + 2 moved to phase 0, timing only)
        if (!(FLAGS & 1024)) {
            if (more1) P::issue(3, SN + 3, c, kt + 1);
            if ((FLAGS & 512) && more2) P::issue(1, S + 2, c, kt + 2);
        }
        P::template read_frags<S + 0, 2>(c, c.rdB, fb0);
        P::template read_frags<S + 1, 2>(c, c.rdB, fb1);
        P::template read_frags<S + 2, 4>(c, c.rdA, fa);
        if (FLAGS & 1024) {
            if (more1) P::issue(3, SN + 3, c, kt + 1);
            if ((FLAGS & 512) && more2) P::issue(1, S + 2, c, kt + 2);
        }
        if (!(FLAGS & 65)) { if (more1) wait_vmcnt<8>(); else wait_vmcnt<0>(); }
        if (!(FLAGS & 128)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P::bar();
        if (FLAGS & 128) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        P::quad(acc[0][0], fa, fb0);
        if (FLAGS & 4096) {                               // timing only: EXTRA ordinary VGPR loads among the MFMAs (2 or, with & 8192, 8 per phase)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < ((FLAGS & 8192) ? 8 : 2); ++u) P::vgpr_load(c, c.junk[u & 3], c.offA[u & 1][(u >> 1) & 1], ((kt + 1) % c.nt) * P::ROWB);
            __builtin_amdgcn_sched_barrier(0);
        }
        P::quad(acc[0][1], fa, fb1);
        P::bar();
        // ---- phase 1
        if (!(FLAGS & 1024) && more2) { P::issue(0, S + 0, c, kt + 2); P::issue(2, S + 1, c, kt + 2); if (!(FLAGS & 512)) P::issue(1, S + 2, c, kt + 2); }
        P::template read_frags<S + 3, 4>(c, c.rdA, fa);
        if ((FLAGS & 1024) && more2) { P::issue(0, S + 0, c, kt + 2); P::issue(2, S + 1, c, kt + 2); if (!(FLAGS & 512)) P::issue(1, S + 2, c, kt + 2); }
        if (!(FLAGS & 65)) { if (more2) wait_vmcnt<8>(); else if (more1) wait_vmcnt<2>(); else wait_vmcnt<0>(); }
        if (!(FLAGS & 128)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P::bar();
        if (FLAGS & 128) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        P::quad(acc[1][1], fa, fb1);
        if (FLAGS & 4096) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < ((FLAGS & 8192) ? 8 : 2); ++u) P::vgpr_load(c, c.junk[u & 3], c.offA[u & 1][(u >> 1) & 1], ((kt + 1) % c.nt) * P::ROWB);
            __builtin_amdgcn_sched_barrier(0);
        }
        P::quad(acc[1][0], fa, fb0);
        P::bar();
    }

    static __device__ __forceinline__ void run(const GemmOperands& p, int a_nrows, int m0, int n0, char* smem, Acc& acc) {
        Ctx c;
        P::setup(c, p, a_nrows, m0, n0, smem);
        const int wr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
        P::issue(0, 0, c, 0, false); P::issue(2, 1, c, 0, false); P::issue(1, 2, c, 0, false); P::issue(3, 3, c, 0, false);
        if (c.nt > 1) { P::issue(0, 4, c, 1, false); P::issue(2, 5, c, 1, false); P::issue(1, 6, c, 1, false); wait_vmcnt<8>(); }
        else wait_vmcnt<2>();
        P::bar();
        if (wr == 1 && !(FLAGS & 4)) P::bar();
        Frag<T> fa[4][CH], fb0[2][CH], fb1[2][CH];
        if (FLAGS & 2) { P::template read_frags<0, 2>(c, c.rdB, fb0, false); P::template read_frags<1, 2>(c, c.rdB, fb1, false); P::template read_frags<2, 4>(c, c.rdA, fa, false); }
        for (int kt = 0; kt < c.nt; kt += 2) {
            ktile<0>(c, kt, acc, fa, fb0, fb1);
            if (kt + 1 < c.nt) ktile<1>(c, kt + 1, acc, fa, fb0, fb1);
        }
        if (wr == 0 && !(FLAGS & 4)) P::bar();
        P::bar();
    }
};

// (GemmP4b -- ALL 24 fragment reads of a k-tile in phase 0 (+32 VGPRs: 250), ALL 8 DMA pieces of k-tile t + 2 in phase 1, one vmcnt(8) per k-tile -- was built on the
//  reading that a DMA piece beside ds_read traffic is what makes a load part long; bit-identical, 1.323 vs 1.309 PF at 8192^3 and equal on the K = 512 shapes: no gain,
//  removed.  profiles/gemm_p8_r06.txt.)
// (A persistent tile-stream form of this loop -- one workgroup per CU walking a tile list with the ring running on across tile boundaries, the next
//  tile's first k-tiles prefetched under the current tile's last, a register-lean epilogue between -- was built in round 6, parity-green, and measured
//  SLOWER than one tile per workgroup on every shape (profiles/gemm_p8_r06.txt: 8192^3 1.13-1.19 vs 1.33-1.35 PF; 36864 x 2736 x 512 with the GEGLU
//  epilogue 211 vs 183 us, the 128 x 128 loop 148): vmcnt counts the epilogue's stores in order with the DMA pieces, so the first counted wait behind an
//  epilogue waits for the stores' acknowledgements; the per-tile offsets no longer fit beside 128 accumulators + 64 fragment registers and have to be
//  rebuilt per DMA piece in the load part of a phase, which is this loop's critical path.  Removed.)

}  // namespace pk
