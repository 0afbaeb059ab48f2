// pk_patch_embed (bf16 operands): the tokenizer's patch embedding up to the Linear -- reference cvivit.py:273-285,
//   Rearrange 'b c (t pt) (h p1) (w p2) -> b t h w (c pt p1 p2)' -> nn.LayerNorm(P) -> nn.Linear(P, dim)
// -- as ONE MFMA GEMM whose A operand is gathered from the (B, C, F, H, W) f32 video on the fly (SURVEY.md 2a "K1").
// Round 2 ran pk_patchify_ln (107 MB read, a 50 MB bf16 patch matrix written) + a GEMM re-reading that matrix; here the patch
// matrix never exists: every workgroup builds its 128-row A tiles in LDS from 128-byte video lines (a k-tile of 64 patch features
// is 64 / pw consecutive lines of the patch), and the LayerNorm is folded into the GEMM:
//     LN(x) W^T + b = rstd * (x' (gamma.W)^T - mean' * s) + t,     x' = x - c,  mean' = mean(x'),  rstd = 1 / sqrt(var(x') + eps),
//     s[n] = sum_k (gamma.W)[n][k],  t[n] = sum_k beta[k] W[n][k] + b[n]
// with c = the mean of the patch's first 32 features: any constant per row leaves LN(x) unchanged, and centring on a value from the
// patch itself keeps the bf16 rounding of the operand relative to the patch's RANGE, not to its mean (a flat, bright patch would
// otherwise lose its whole signal in the rounding of x ~ mean), and makes the one-pass variance well conditioned (measured on the
// BASELINE geometry: same bf16-vs-f32 error as rounding AFTER the LayerNorm, rms 2.4e-3).  The statistics are f32 sums of the unrounded
// x' taken as the tile is converted; operands are bf16(x') and bf16(gamma.W), accumulation f32.
// Tile: 128 patch rows x 128 (64 for narrow outputs) columns, 8 waves as 4 x 2 (wave tile 32 x 64), k-tile 64.  W tiles arrive by
// LDS-DMA into a ring of 4 (issued 3 k-tiles ahead); A tiles are register-staged: buffer_load_dwordx4 issued through inline asm THREE
// k-tiles ahead (hipcc would drain the whole queue with vmcnt(0) at the first use of an ordinary load), centred / summed / converted one
// k-tile ahead and written with ds_write_b64 into a 2-stage copy of the XOR-swizzled image the DMA kernels use.  Every wait is the same
// counted vmcnt; one barrier per k-tile.  Both frame groups (first frame: pt = 1; the rest: pt = temporal patch size) run in one launch.
// What bounds it (MI355X, B = 8: 4096 + 512 rows, P = 6144 / 3072, dim 512; profiles/patch_embed_r03.txt): the L2 -> CU fill path.  The f32
// A operand is re-read once per column tile (4 B per feature against 2 B for W), and a CU sustains ~25 B/clk of such fills: 64-column
// tiles move 1.03 GB through it (93-99 us), 128-column tiles 0.6 GB on 144 workgroups (77-79 us); a deeper register pipeline (2 -> 3
// k-tiles) and a per-m-tile rotation of the k walk (PK_PATCH_ROT, against HBM channel pile-ups) changed nothing.  Against round 2's
// pk_patchify_ln + GEMM (28 + 5.5 + 2 x 33 us) that is 77 us and three launches fewer: the encode step 0.85 -> 0.80 ms (same-box A/B).
// Roofline: 2 * rows * dim * P flops on the matrix cores (335 TFLOP/s reached); the f32 video is read once from HBM.
#include <cstdlib>
#include "gemm_dma.hpp"

// one frame group of pk_patch_embed_finish_groups (mirrors include/phenaki_hip.h)
extern "C" {
typedef struct {
    const float* part; const float* stats; int nslices, rows, K;
    const float* s; const float* t; float eps1;
    const float* gamma2; const float* beta2; float eps2;
    int remap_in, remap_out, remap_off;
} pk_patch_finish_group;
}

namespace pk {

struct PatchGroup {
    const void* W;            // [N][ldw] bf16: gamma (.) W of this group's Linear, zero-padded along K to the k-tile
    const float* s;           // [N] row sums of that (rounded) operand
    const float* t;           // [N] W beta + Linear bias
    float* out;               // [rows][ldo] f32: LayerNorm(P) -> Linear output (the caller applies the LayerNorm(dim) that follows)
    int ldw, K;               // K = C * pt * ph * pw
    int f0, nt, pt;           // first frame of the group, temporal patches per video, frames per patch
    int rows;                 // B * nt * nh * nw
    int mtiles;               // ceil(rows / 128)
};

struct PatchEmbedArgs {
    const float* video;
    int B, C, F, H, W, ph, pw, nh, nw;
    int N, ldo;
    float eps;
    uint32_t video_bytes;
    int ngroups;
    int rot_mult;             // k-tile rotation step per m-tile (0: every workgroup starts at k = 0)
    PatchGroup g[2];
};

constexpr int PE_BM = 128, PE_WAVES = 8, PE_ASTAGE = PE_BM * 128, PE_WRING = 4;
// TN = MFMA column blocks per wave: BN = 32 * TN output columns per workgroup (wave grid 4 x 2, wave tile 32 x 16 TN)
template <int TN> struct PeGeom {
    static constexpr int BN = 32 * TN, WSTAGE = BN * 128, IW = BN / 64, SMEM = 2 * PE_ASTAGE + PE_WRING * WSTAGE, WAITN = 2 * (IW + 4);
};

typedef __attribute__((address_space(3))) void* pe_lds_ptr;

// 16-byte buffer load the compiler does not see as a memory operation: no automatic s_waitcnt (it would be vmcnt(0), draining the W
// tile's LDS-DMA and the prefetch two k-tiles ahead).  Every use is preceded by pe_wait<N>() naming the destination registers.
__device__ __forceinline__ void pe_load16(f32x4& dst, uint32_t voff, const u32x4& rsrc, uint32_t soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void pe_wait(f32x4 (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%[cnt])" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : [cnt] "n"(N) : "memory");
}

// (the body is a __device__ template called from two plain __global__ kernels: a __global__ TEMPLATE whose body holds AMDGPU inline asm is
// silently not instantiated by the host pass of hipcc 7.2 -- the launch stub then stays an undefined symbol)
template <int TN>
__device__ __forceinline__ void patch_embed_body(const PatchEmbedArgs& a, char* smem) {
    constexpr int PE_BN = PeGeom<TN>::BN, PE_WSTAGE = PeGeom<TN>::WSTAGE, IW = PeGeom<TN>::IW, WAITN = PeGeom<TN>::WAITN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- tile map: the m-tiles of all groups (long-K group first) form one list; XCD x owns a contiguous chunk of it and walks every
    // n-tile (m fastest), so the 8 column tiles that read the same video lines run on one XCD and share them through its L2
    const int NT = (a.N + PE_BN - 1) / PE_BN;
    const int MT = a.g[0].mtiles + (a.ngroups > 1 ? a.g[1].mtiles : 0), cmax = (MT + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mstart = xcd * MT / 8, mcount = (xcd + 1) * MT / 8 - mstart;
    const int ml = idx % cmax;
    if (ml >= mcount) return;
    int mt = mstart + ml;
    const int n0 = (idx / cmax) * PE_BN;
    const int gi = (a.ngroups > 1 && mt >= a.g[0].mtiles) ? 1 : 0;
    if (gi) mt -= a.g[0].mtiles;
    const PatchGroup& G = a.g[gi];
    const int m0 = mt * PE_BM;
    const int K = G.K, nt = K >> 6;                         // host: K % 192 == 0 (the pipeline is unrolled by its 3 register sets)
    const int rot = (mt * a.rot_mult) % nt;                          // this m-tile's first k-tile (shared by its column tiles)

    // ---- A side: lane -> (octet o = 8 lanes sharing a row, 16-byte piece p); instruction (hrow, half) covers rows wave*16 + hrow*8 + perm(o)
    // of the tile and features [kt*64 + half*32, +32).  perm makes the two rows of a 16-lane ds_write group differ in bit 2 (disjoint banks).
    const int o = lane >> 3, p = lane & 7;
    const int rr = ((o & 1) << 2) | (o >> 1);               // 0,4,1,5,2,6,3,7
    const int pw = a.pw;
    // element e = p*4 of the half: pw >= 32: same line, x = e;  pw < 32: line e / pw, x = e % pw  (host: pw a power of two, 8..128)
    const int dline = pw < 32 ? (p * 4) / pw : 0, xin = pw < 32 ? (p * 4) % pw : p * 4;
    const int per_b = G.nt * a.nh * a.nw;
    uint32_t rowoff[2];                                     // byte offset of (row's patch origin + this lane's piece) in the video
    float cen[2];
#pragma unroll
    for (int hrow = 0; hrow < 2; ++hrow) {
        const int r = m0 + wave * 16 + hrow * 8 + rr;
        const bool ok = r < G.rows;
        const int rc = ok ? r : 0;
        const int b = rc / per_b, rem = rc - b * per_b;
        const int tt = rem / (a.nh * a.nw), hw = rem - tt * (a.nh * a.nw);
        const int hh = hw / a.nw, ww = hw - hh * a.nw;
        const uint32_t org = (((uint32_t)(b * a.C) * a.F + (G.f0 + tt * G.pt)) * a.H + hh * a.ph) * a.W + ww * a.pw;      // elements
        // centre: mean of the row's first 32 features (this lane's piece of them, folded over the octet)
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(a.video + org + (uint32_t)dline * a.W + xin);
        float cs = (c4[0] + c4[1]) + (c4[2] + c4[3]);
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) cs += __shfl_xor(cs, off, 64);
        cen[hrow] = ok ? cs * (1.0f / 32.0f) : 0.f;
        rowoff[hrow] = ok ? (org + (uint32_t)dline * a.W + xin) * 4u : a.video_bytes;     // invalid rows: out of bounds -> the loads return 0
    }
    const uintptr_t vb = reinterpret_cast<uintptr_t>(a.video);
    const u32x4 rsrc = u32x4{(uint32_t)vb, (uint32_t)(vb >> 32) & 0xFFFFu, a.video_bytes, 0x00020000u};

    // scalar walk over the halves (32 features each) from the rotated start, wrapping at K: line (cc, dt, y) of the patch and x0 inside
    // it.  The pipeline below issues its loads UNCONDITIONALLY (straight-line asm, one wait count for every iteration -- with conditional
    // issues hipcc merged the in-flight registers of the two paths with v_mov copies placed BEFORE the wait), so the k-tiles "after the
    // end" simply wrap around to valid addresses and are never converted.
    const int lph = pw < 32 ? 32 / pw : 1;                  // lines per half
    int hx0, hy, hdt, hcc;                                  // state of the NEXT half to be issued
    {
        const int e0 = rot * 64, line0 = e0 / pw;
        hx0 = e0 - line0 * pw;
        hy = line0 % a.ph;
        const int r2 = line0 / a.ph;
        hdt = r2 % G.pt; hcc = r2 / G.pt;
    }
    auto next_soff = [&]() {
        const uint32_t so = ((((uint32_t)hcc * a.F + hdt) * a.H + hy) * a.W + hx0) * 4u;
        hx0 += 32;
        if (hx0 >= pw) {
            hx0 = 0; hy += lph;
            if (hy >= a.ph) { hy = 0; if (++hdt == G.pt) { hdt = 0; if (++hcc == a.C) hcc = 0; } }
        }
        return so;
    };
    auto issue_a = [&](f32x4 (&r)[4]) {                     // one k-tile: 2 halves x 2 row groups
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t so = next_soff();
            pe_load16(r[half * 2 + 0], rowoff[0], rsrc, so);
            pe_load16(r[half * 2 + 1], rowoff[1], rsrc, so);
        }
    };

    // ---- W side (gemm_dma.hpp's image): one 1 KiB DMA piece per wave per k-tile = 8 rows x 128 B, slot swizzle on the source address
    char* const wring = smem + 2 * PE_ASTAGE;
    const int lrow = lane >> 3, lslot = lane & 7;
    const uint32_t bytesW = (uint32_t)a.N * (uint32_t)G.ldw * 2u;
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(G.W), 0, bytesW, 0x00020000);
    uint32_t offW[IW];
#pragma unroll
    for (int i = 0; i < IW; ++i) {
        const int gn = n0 + (wave * IW + i) * 8 + lrow;
        offW[i] = gn < a.N ? (uint32_t)gn * (uint32_t)G.ldw * 2u + (uint32_t)((lslot ^ (lrow & 7)) * 16) : bytesW;
    }
    int wk = rot;                                           // k-tile index of the NEXT W tile to be issued (wraps like the A walk)
    auto issue_w = [&](int j) {                             // j: position in this workgroup's (rotated) k-tile sequence
#pragma unroll
        for (int i = 0; i < IW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (pe_lds_ptr)(wring + (j & (PE_WRING - 1)) * PE_WSTAGE + (wave * IW + i) * 1024), 16, offW[i], wk * 128, 0, 0);
        if (++wk == nt) wk = 0;
    };

    float rs[2] = {0.f, 0.f}, rq[2] = {0.f, 0.f};
    // centre, accumulate the row statistics, round to bf16, store into A stage `stage`
    auto convert_store = [&](const f32x4 (&r)[4], int stage) {
        char* at = smem + stage * PE_ASTAGE;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int hrow = 0; hrow < 2; ++hrow) {
                f32x4 x = r[half * 2 + hrow];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] -= cen[hrow];
                    rs[hrow] += x[e];
                    rq[hrow] = fmaf(x[e], x[e], rq[hrow]);
                }
                const int row = wave * 16 + hrow * 8 + rr;
                const int slot = (half * 4 + (p >> 1)) ^ (row & 7);
                *reinterpret_cast<u32x2*>(at + row * 128 + (slot << 4) + ((p & 1) << 3)) = u32x2{pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
            }
    };

    const int g = lane >> 4, lr = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    auto compute = [&](int j) {
        const char* at = smem + (j & 1) * PE_ASTAGE;
        const char* wt = wring + (j & (PE_WRING - 1)) * PE_WSTAGE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Frag<bf16> fa[2], fw[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) lds_frag(fa[i], at, wm * 32 + i * 16 + lr, c, g);
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) lds_frag(fw[jj], wt, wn * 16 * TN + jj * 16 + lr, c, g);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) acc[i][jj] = mma(fw[jj], fa[i], acc[i][jj]);
        }
    };

    // ---- pipeline (positions j = 0 .. nt-1 in the rotated k-tile sequence).  Iteration j issues [W(j+3) x IW] [A(j+4) x4] and ends with a
    // wait for W(j+1) and A(j+2), both issued in iteration j-2; VMEM operations of a wave complete in issue order, so allowing the
    // 2 (IW + 4) most recent ones (iterations j-1 and j) to stay in flight retires exactly what is needed.  The prologue issues in the same order.
    f32x4 r0[4], r1[4], r2[4];                              // A(j) lives in set j % 3 (A(0) borrows r0 before A(3))
    issue_a(r0);                                            // A(0)
    issue_w(0); issue_a(r1);                                // "iteration -3"
    issue_w(1); issue_a(r2);                                // "iteration -2"
    pe_wait<WAITN>(r0);
    __builtin_amdgcn_sched_barrier(0);
    convert_store(r0, 0);
    issue_w(2); issue_a(r0);                                // "iteration -1": A(3)
    pe_wait<WAITN>(r1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // one iteration: `nxt_regs` hold A(j+1) (landed: previous wait) and are refilled with A(j+4); `wait_regs` hold A(j+2)
    auto iteration = [&](int j, f32x4 (&nxt_regs)[4], f32x4 (&wait_regs)[4]) {
        if (j + 1 < nt) convert_store(nxt_regs, (j + 1) & 1);    // that A stage is free: every wave passed the barrier after reading tile j-1
        issue_w(j + 3);                                       // ring slot (j+3) % 4 held tile j-1
        issue_a(nxt_regs);
        compute(j);
        pe_wait<WAITN>(wait_regs);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int j = 0; j < nt; j += 3) {                       // nt % 3 == 0 (host)
        iteration(j, r1, r2);
        iteration(j + 1, r2, r0);
        iteration(j + 2, r0, r1);
    }
    pe_wait<0>(r0);                                         // the loads issued past the end must land before their registers are reused
    pe_wait<0>(r1);
    pe_wait<0>(r2);
    __builtin_amdgcn_sched_barrier(0);

    // ---- row statistics: fold the 8 lanes of an octet, hand (mean', rstd) of the tile's 128 rows over through the dead ring
    float2* st = reinterpret_cast<float2*>(smem);
#pragma unroll
    for (int hrow = 0; hrow < 2; ++hrow) {
        float s = rs[hrow], q = rq[hrow];
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
        const float inv_k = 1.0f / (float)K;
        const float mean = s * inv_k;
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(q * inv_k - mean * mean, 0.f) + a.eps);
        if (p == 0) st[wave * 16 + hrow * 8 + rr] = float2{mean, rstd};
    }
    __syncthreads();
    // ---- epilogue: out = rstd * (acc - mean' * s) + t ; lane holds 4 consecutive columns of rows (wm*32 + i*16 + lr)
    f32x4 s4[TN], t4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = min(n0 + wn * 16 * TN + j * 16 + g * 4, a.N - 4);       // N % 4 == 0 (host); clamped loads are never stored
        s4[j] = *reinterpret_cast<const f32x4*>(G.s + n);
        t4[j] = *reinterpret_cast<const f32x4*>(G.t + n);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = wm * 32 + i * 16 + lr, m = m0 + rl;
        const float2 ms = st[rl];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 16 * TN + j * 16 + g * 4;
            if (m < G.rows && n < a.N) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = ms.y * (acc[i][j][r] - ms.x * s4[j][r]) + t4[j][r];
                store4(G.out + (size_t)m * a.ldo + n, v);
            }
        }
    }
}

__global__ __launch_bounds__(64 * PE_WAVES) __attribute__((amdgpu_waves_per_eu(4, 8)))       // 64-column tiles: 2 workgroups per CU
void patch_embed_kernel_n64(const PatchEmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    patch_embed_body<2>(a, smem);
}
__global__ __launch_bounds__(64 * PE_WAVES) __attribute__((amdgpu_waves_per_eu(2, 8)))       // 128-column tiles: 96 KB of LDS, 1 workgroup per CU
void patch_embed_kernel_n128(const PatchEmbedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    patch_embed_body<4>(a, smem);
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// Round 4: the same product as ROW PANELS x ALL COLUMNS x K-SLICES (pk_patch_embed_splitk + pk_patch_embed_finish).
// The 128 x 128 tiling above re-reads the f32 video lines once per column tile (4 x 107 MB through the CU's 64 B/clk L1 path at B = 8) and has
// only 144 workgroups; its PMC fetch was 1.58 x the algorithmic bytes.  Here a workgroup owns 128 patch rows x ALL N <= 512 output columns x one
// K-slice of PW_KS = 1024 features: the video is read exactly once (each line by one workgroup), W (bf16, L2-resident) streams through a 2-stage
// LDS-DMA ring of 64 KB k-tiles, 8 waves as 2 x 4 hold 64 x 128 accumulator tiles, and the 6 (P = 6144) / 3 (P = 3072) slices of a row panel
// bring 32 x 6 + 4 x 3 = 204 workgroups to the 256 CUs in ONE round.  Per k-tile a workgroup moves 32 KB of f32 lines + 64 KB of W for
// 8.4 MFLOP (87 flop per L1 byte against 65 for 128 x 128 tiles).  Every slice writes its raw partial product and the partial (sum, sum of
// squares) of its centred rows; pk_patch_embed_finish adds the slices in index order (deterministic), applies the folded LayerNorm(P), the
// Linear's bias, and the LayerNorm(dim) that follows -- the launch that used to be pk_layernorm -- writing the token rows (f32 + bf16 copy).
constexpr int PW_BN = 512, PW_WSTAGE = PW_BN * 128, PW_KS = 1024, PW_SMEM = 2 * PE_ASTAGE + 2 * PW_WSTAGE;      // 160 KB: the whole LDS of a CU

struct PatchWideGroup {
    const void* W; int ldw, K;
    int f0, nt, pt, rows, mtiles, nslices;
    float* part;              // [nslices][rows][N] f32 raw partial products x' (gamma.W)^T
    float* stats;             // [nslices][rows][2] f32 partial (sum x', sum x'^2)
};
struct PatchWideArgs {
    const float* video;
    int B, C, F, H, W, ph, pw, nh, nw;
    int N;
    uint32_t video_bytes;
    int ngroups;
    int dbg;                  // diagnostics (PK_PATCH_DBG; results are WRONG when set): 1 = no MFMAs, 2 = no W stream, 4 = no A conversion
    PatchWideGroup g[2];
};

__device__ __forceinline__ void patch_embed_wide_body(const PatchWideArgs& a, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- work list: (group, slice, m-tile) in SLICE-MAJOR order, long-K group first; XCD x (workgroup b runs on XCD b % 8: speed only) owns a
    // contiguous chunk of it, i.e. the row panels of one or two K-slices: their 1 MB of W is fetched over the fabric once per XCD and then
    // served by its L2, instead of every workgroup streaming its own slice from the Infinity Cache (measured: 204 MB of W fabric traffic, 48.7 us)
    const int n0w = a.g[0].mtiles * a.g[0].nslices;
    const int nwork = n0w + (a.ngroups > 1 ? a.g[1].mtiles * a.g[1].nslices : 0);
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int wstart = xcd * nwork / 8, wcount = (xcd + 1) * nwork / 8 - wstart;
    if (idx >= wcount) return;
    int w = wstart + idx;
    const int gi = (a.ngroups > 1 && w >= n0w) ? 1 : 0;
    if (gi) w -= n0w;
    const PatchWideGroup& G = a.g[gi];
    const int slice = w / G.mtiles, mt = w - slice * G.mtiles;
    const int m0 = mt * PE_BM;
    const int K = G.K, ntk = K >> 6;                         // k-tiles of the whole row
    const int kt0 = slice * (PW_KS >> 6);
    const int nts = min(PW_KS >> 6, ntk - kt0);              // k-tiles of this slice (host: K % 64 == 0)

    // ---- A side (as in patch_embed_body): lane -> (octet o = 8 lanes sharing a row, 16-byte piece p)
    const int o = lane >> 3, p = lane & 7;
    const int rr = ((o & 1) << 2) | (o >> 1);
    const int pw = a.pw;
    const int dline = pw < 32 ? (p * 4) / pw : 0, xin = pw < 32 ? (p * 4) % pw : p * 4;
    const int per_b = G.nt * a.nh * a.nw;
    uint32_t rowoff[2];
    float cen[2];
#pragma unroll
    for (int hrow = 0; hrow < 2; ++hrow) {
        const int r = m0 + wave * 16 + hrow * 8 + rr;
        const bool ok = r < G.rows;
        const int rc = ok ? r : 0;
        const int b = rc / per_b, rem = rc - b * per_b;
        const int tt = rem / (a.nh * a.nw), hw = rem - tt * (a.nh * a.nw);
        const int hh = hw / a.nw, ww = hw - hh * a.nw;
        const uint32_t org = (((uint32_t)(b * a.C) * a.F + (G.f0 + tt * G.pt)) * a.H + hh * a.ph) * a.W + ww * a.pw;
        // the SAME centre for every slice of a row: the mean of the patch's first 32 features
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(a.video + org + (uint32_t)dline * a.W + xin);
        float cs = (c4[0] + c4[1]) + (c4[2] + c4[3]);
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) cs += __shfl_xor(cs, off, 64);
        cen[hrow] = ok ? cs * (1.0f / 32.0f) : 0.f;
        rowoff[hrow] = ok ? (org + (uint32_t)dline * a.W + xin) * 4u : a.video_bytes;
    }
    const uintptr_t vb = reinterpret_cast<uintptr_t>(a.video);
    const u32x4 rsrc = u32x4{(uint32_t)vb, (uint32_t)(vb >> 32) & 0xFFFFu, a.video_bytes, 0x00020000u};
    const int lph = pw < 32 ? 32 / pw : 1;
    int hx0, hy, hdt, hcc;
    {
        const int e0 = kt0 * 64, line0 = e0 / pw;
        hx0 = e0 - line0 * pw;
        hy = line0 % a.ph;
        const int r2 = line0 / a.ph;
        hdt = r2 % G.pt; hcc = r2 / G.pt;
    }
    auto next_soff = [&]() {
        const uint32_t so = ((((uint32_t)hcc * a.F + hdt) * a.H + hy) * a.W + hx0) * 4u;
        hx0 += 32;
        if (hx0 >= pw) {
            hx0 = 0; hy += lph;
            if (hy >= a.ph) { hy = 0; if (++hdt == G.pt) { hdt = 0; if (++hcc == a.C) hcc = 0; } }
        }
        return so;
    };
    auto issue_a = [&](f32x4 (&r)[4]) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t so = next_soff();
            pe_load16(r[half * 2 + 0], rowoff[0], rsrc, so);
            pe_load16(r[half * 2 + 1], rowoff[1], rsrc, so);
        }
    };

    // ---- W side: a k-tile of all N columns = N / 8 DMA pieces, 8 per wave (rows past N read as zeros through the descriptor)
    char* const wring = smem + 2 * PE_ASTAGE;
    const int lrow = lane >> 3, lslot = lane & 7;
    const uint32_t bytesW = (uint32_t)a.N * (uint32_t)G.ldw * 2u;
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(G.W), 0, bytesW, 0x00020000);
    uint32_t offW[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gn = (wave * 8 + i) * 8 + lrow;
        offW[i] = gn < a.N ? (uint32_t)gn * (uint32_t)G.ldw * 2u + (uint32_t)((lslot ^ (lrow & 7)) * 16) : bytesW;
    }
    int wk = kt0;
    auto issue_w = [&](int j) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (pe_lds_ptr)(wring + (j & 1) * PW_WSTAGE + (wave * 8 + i) * 1024), 16, offW[i], wk * 128, 0, 0);
        if (++wk == ntk) wk = 0;
    };

    float rs[2] = {0.f, 0.f}, rq[2] = {0.f, 0.f};
    auto convert_store = [&](const f32x4 (&r)[4], int stage) {
        char* at = smem + stage * PE_ASTAGE;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int hrow = 0; hrow < 2; ++hrow) {
                f32x4 x = r[half * 2 + hrow];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] -= cen[hrow];
                    rs[hrow] += x[e];
                    rq[hrow] = fmaf(x[e], x[e], rq[hrow]);
                }
                const int row = wave * 16 + hrow * 8 + rr;
                const int slot = (half * 4 + (p >> 1)) ^ (row & 7);
                *reinterpret_cast<u32x2*>(at + row * 128 + (slot << 4) + ((p & 1) << 3)) = u32x2{pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
            }
    };

    const int g = lane >> 4, lr = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;                // 2 x 4 waves: wave tile 64 rows x 128 columns
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    auto compute_half = [&](int j, int c) {                 // fragment chunk c (32 of the k-tile's 64 features): 32 MFMAs per wave
        const char* at = smem + (j & 1) * PE_ASTAGE;
        const char* wt = wring + (j & 1) * PW_WSTAGE;
        Frag<bf16> fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_frag(fa[i], at, wm * 64 + i * 16 + lr, c, g);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            Frag<bf16> fw;
            lds_frag(fw, wt, wn * 128 + jj * 16 + lr, c, g);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][jj] = mma(fw, fa[i], acc[i][jj]);
        }
    };
    // (measured and removed: waves w / w + 4 of a SIMD issuing their W pieces half an iteration apart -- 53.0 vs 52.6 us, no effect)

    // ---- pipeline.  A(j) lives in register set j % 3, issued three k-tiles ahead, converted into LDS stage (j & 1) one k-tile ahead; W(j) goes by
    // LDS-DMA into ring slot (j & 1), issued one k-tile ahead (the slot is free once every wave has passed the barrier that ended iteration j - 2's
    // successor).  Iteration j issues [W(j+1) x 8][A(j+4) x 4] and ends waiting for all but the 4 most recent VMEM operations: that retires
    // W(j+1) and every older A set.  Loads past the slice wrap to valid addresses and are never converted.
    f32x4 r0[4], r1[4], r2[4];
    issue_a(r0);                                            // A(0)
    issue_w(0);                                             // W(0)
    issue_a(r1);                                            // A(1)
    issue_a(r2);                                            // A(2)
    pe_wait<8>(r0);                                         // A(0) and W(0) landed (A(1), A(2) may fly)
    __builtin_amdgcn_sched_barrier(0);
    convert_store(r0, 0);
    issue_a(r0);                                            // A(3)
    pe_wait<8>(r1);                                         // A(1) landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    auto iteration = [&](int j, f32x4 (&nxt_regs)[4], f32x4 (&wait_regs)[4]) {
        if (j + 1 < nts && !(a.dbg & 4)) convert_store(nxt_regs, (j + 1) & 1);   // A stage (j+1)&1 was last read in iteration j-1: every wave is past that barrier
        if (!(a.dbg & 2)) issue_w(j + 1);                         // W slot (j+1)&1 was last read in iteration j-1 likewise
        issue_a(nxt_regs);                                        // A(j+4): unconditional (see the toolchain note above)
        if (j < nts && !(a.dbg & 1)) { compute_half(j, 0); compute_half(j, 1); }
        pe_wait<4>(wait_regs);                                    // W(j+1) landed; wait_regs = A(j+2) (older) landed too
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int j = 0; j < nts; j += 3) {                      // the last round may run up to 2 iterations past the slice: they only issue and wait
        iteration(j, r1, r2);
        iteration(j + 1, r2, r0);
        iteration(j + 2, r0, r1);
    }
    pe_wait<0>(r0);
    pe_wait<0>(r1);
    pe_wait<0>(r2);
    __builtin_amdgcn_sched_barrier(0);

    // ---- this slice's partial statistics of the wave's 16 rows (fold the 8 lanes of an octet)
#pragma unroll
    for (int hrow = 0; hrow < 2; ++hrow) {
        float s = rs[hrow], q = rq[hrow];
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) { s += __shfl_xor(s, off, 64); q += __shfl_xor(q, off, 64); }
        const int m = m0 + wave * 16 + hrow * 8 + rr;
        if (p == 0 && m < G.rows) reinterpret_cast<float2*>(G.stats)[(size_t)slice * G.rows + m] = float2{s, q};
    }
    // ---- raw partial product: lane holds 4 consecutive columns of rows wm*64 + i*16 + lr
    float* part = G.part + (size_t)slice * G.rows * a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + lr;
        if (m >= G.rows) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = wn * 128 + j * 16 + g * 4;
            if (n < a.N) store4(part + (size_t)m * a.N + n, acc[i][j]);
        }
    }
}

__global__ __launch_bounds__(64 * PE_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))       // the whole LDS: one workgroup per CU
void patch_embed_wide_kernel(const PatchWideArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    patch_embed_wide_body(a, smem);
}

// pk_patch_embed_finish: one wave per token row.  y = rstd (sum_s part_s - mean' s) + t over the row's P features (statistics = the slices'
// partial sums), then LayerNorm(dim) of y (gamma2, beta2, eps2) -> out2 (f32) and / or out (bf16), rows remapped as pk_layernorm's remap does.
struct PatchFinishArgs {
    const float* part; const float* stats; int nslices, rows, N, K;
    const float* s; const float* t; float eps1;
    const float* gamma2; const float* beta2; float eps2;
    float* out2; bf16* out; int ldo2, ldo;
    int remap_in, remap_out, remap_off;
};
__device__ __forceinline__ void patch_embed_finish_rows(const PatchFinishArgs& a, int block) {
    const int lane = threadIdx.x & 63;
    const int row = block * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    float su = 0.f, sq = 0.f;
    for (int sl = 0; sl < a.nslices; ++sl) {                 // slices in index order: deterministic
        const float2 st = reinterpret_cast<const float2*>(a.stats)[(size_t)sl * a.rows + row];
        su += st.x; sq += st.y;
    }
    const float inv_k = 1.0f / (float)a.K;
    const float mean = su * inv_k;
    const float rstd = 1.0f / sqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + a.eps1);
    // N <= 512: up to two 4-column pieces per lane (columns lane*4 + q*256)
    f32x4 y[2];
    float ls = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = lane * 4 + q * 256;
        y[q] = f32x4{0, 0, 0, 0};
        if (n < a.N) {
            f32x4 acc = f32x4{0, 0, 0, 0};
            for (int sl = 0; sl < a.nslices; ++sl) acc += *reinterpret_cast<const f32x4*>(a.part + ((size_t)sl * a.rows + row) * a.N + n);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(a.s + n), t4 = *reinterpret_cast<const f32x4*>(a.t + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) { y[q][r] = rstd * (acc[r] - mean * s4[r]) + t4[r]; ls += y[q][r]; }
        }
    }
    const float m2 = wave_sum(ls) / (float)a.N;
    float lq = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (lane * 4 + q * 256 < a.N)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = y[q][r] - m2; lq = fmaf(d, d, lq); }
    const float r2 = 1.0f / sqrtf(wave_sum(lq) / (float)a.N + a.eps2);
    const int orow = a.remap_in > 0 ? (row / a.remap_in) * a.remap_out + a.remap_off + row % a.remap_in : row;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int n = lane * 4 + q * 256;
        if (n >= a.N) continue;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.gamma2 + n), b4 = *reinterpret_cast<const f32x4*>(a.beta2 + n);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (y[q][r] - m2) * r2 * g4[r] + b4[r];
        if (a.out2) store4(a.out2 + (size_t)orow * a.ldo2 + n, v);
        if (a.out) store4(a.out + (size_t)orow * a.ldo + n, v);
    }
}
__global__ __launch_bounds__(256) void patch_embed_finish_kernel(const PatchFinishArgs a) { patch_embed_finish_rows(a, blockIdx.x); }
// both frame groups of a video batch in ONE launch (round 5: the second finish launch was 5-10 us of ramp for 512 rows): blocks [0, blocks0) serve group 0
struct PatchFinishPair { PatchFinishArgs g[2]; int blocks0; };
__global__ __launch_bounds__(256) void patch_embed_finish_pair_kernel(const PatchFinishPair p) {
    if ((int)blockIdx.x < p.blocks0) patch_embed_finish_rows(p.g[0], blockIdx.x);
    else patch_embed_finish_rows(p.g[1], blockIdx.x - p.blocks0);
}

}  // namespace pk
using namespace pk;

// video (B, C, F, H, W) f32 -> per frame group the LayerNorm(P) + Linear(P, N) output rows [(b, tt, hh, ww)][N] f32.
// Group gi (0: frames [f0_0, f0_0 + nt_0 * pt_0), ...; ngroups 1 or 2): W_gi [N][ldw_gi] bf16 = gamma (.) W zero-padded along K to 64,
// s_gi / t_gi [N] f32 (see the file header), out_gi [B * nt_gi * nh * nw][ldo] f32.
extern "C" int pk_patch_embed(const float* video, int B, int C, int F, int H, int W, int ph, int pw, int N, float eps, int ngroups,
                              const void* W0, int ldw0, const float* s0, const float* t0, float* out0, int f00, int nt0, int pt0,
                              const void* W1, int ldw1, const float* s1, const float* t1, float* out1, int f01, int nt1, int pt1,
                              int ldo, void* stream) {
    if (!video || B <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || N <= 0 || (ngroups != 1 && ngroups != 2)) return PK_EINVAL;
    if (H % ph || W % pw || (pw & (pw - 1)) || pw < 8 || pw > 128 || (W & 3) || (N & 3) || (ldo & 3)) return PK_EINVAL;
    if (pw < 32 && ph % (32 / pw)) return PK_EINVAL;                      // a 32-feature half must stay inside one (c, dt) plane
    const size_t vbytes = (size_t)B * C * F * H * W * 4;
    if (vbytes >= 0xFFFFFFF0ull) return PK_EINVAL;                        // 32-bit buffer offsets
    auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
    PatchEmbedArgs a;
    a.video = video; a.B = B; a.C = C; a.F = F; a.H = H; a.W = W; a.ph = ph; a.pw = pw; a.nh = H / ph; a.nw = W / pw;
    a.N = N; a.ldo = ldo; a.eps = eps; a.video_bytes = (uint32_t)vbytes; a.ngroups = ngroups;
    static const int rot_env = [] { const char* e = getenv("PK_PATCH_ROT"); return e ? atoi(e) : 0; }();      // tuning knob (measured neutral: off)
    a.rot_mult = rot_env;
    const void* Ws[2] = {W0, W1}; const int ldws[2] = {ldw0, ldw1}; const float* ss[2] = {s0, s1}; const float* ts[2] = {t0, t1};
    float* outs[2] = {out0, out1}; const int f0s[2] = {f00, f01}, nts[2] = {nt0, nt1}, pts[2] = {pt0, pt1};
    int MT = 0;
    for (int gi = 0; gi < ngroups; ++gi) {
        PatchGroup& G = a.g[gi];
        if (!Ws[gi] || !ss[gi] || !ts[gi] || !outs[gi] || nts[gi] <= 0 || pts[gi] <= 0 || f0s[gi] < 0 || f0s[gi] + nts[gi] * pts[gi] > F) return PK_EINVAL;
        if (mis(video) || mis(Ws[gi]) || mis(ss[gi]) || mis(ts[gi]) || mis(outs[gi])) return PK_EALIGN;
        G.W = Ws[gi]; G.s = ss[gi]; G.t = ts[gi]; G.out = outs[gi]; G.ldw = ldws[gi];
        G.K = C * pts[gi] * ph * pw;
        if (G.K % 192 || G.ldw < G.K || (G.ldw & 7) || (size_t)N * G.ldw * 2 >= 0xFFFFFFF0ull) return PK_EINVAL;
        G.f0 = f0s[gi]; G.nt = nts[gi]; G.pt = pts[gi];
        G.rows = B * nts[gi] * a.nh * a.nw;
        G.mtiles = (G.rows + PE_BM - 1) / PE_BM;
        MT += G.mtiles;
    }
    if (ngroups == 1) a.g[1] = a.g[0];
    // 128-column tiles whenever the output is that wide: the A operand (f32 video lines, 4 bytes per feature) is re-read once per column
    // tile and these shapes are bound by the L2 -> CU traffic (B = 8: 805 MB of A with 64-column tiles = 97 us at ~11 TB/s)
    static const int tn_env = [] { const char* e = getenv("PK_PATCH_TN"); return e ? atoi(e) : 0; }();      // tuning knob: 2 / 4
    const bool wide = tn_env ? tn_env == 4 : N >= 128;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (wide) {
        static bool attr_set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PK_ELAUNCH;
        if (!attr_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_embed_kernel_n128), hipFuncAttributeMaxDynamicSharedMemorySize, PeGeom<4>::SMEM) != hipSuccess) return PK_ELAUNCH;
            attr_set[dev] = true;
        }
        const int NT = (N + PeGeom<4>::BN - 1) / PeGeom<4>::BN;
        hipLaunchKernelGGL(patch_embed_kernel_n128, dim3(8 * ((MT + 7) / 8) * NT), dim3(64 * PE_WAVES), PeGeom<4>::SMEM, st, a);
    } else {
        const int NT = (N + PeGeom<2>::BN - 1) / PeGeom<2>::BN;
        hipLaunchKernelGGL(patch_embed_kernel_n64, dim3(8 * ((MT + 7) / 8) * NT), dim3(64 * PE_WAVES), PeGeom<2>::SMEM, st, a);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}


// K-slice length of pk_patch_embed_splitk (features per slice); the caller sizes part / stats with pk_patch_embed_slices(K)
extern "C" int pk_patch_embed_slices(int K) { return K <= 0 ? 0 : (K + PW_KS - 1) / PW_KS; }

// Row panels x all columns x K-slices (see the kernel): per group gi the raw partial products part_gi [slices_gi][rows_gi][N] f32 and the partial
// row statistics stats_gi [slices_gi][rows_gi][2] f32, slices_gi = pk_patch_embed_slices(C * pt_gi * ph * pw); finish with pk_patch_embed_finish.
// W_gi [N][ldw_gi] bf16 = gamma (.) W zero-padded along K to 64.  N <= 512, N % 4 == 0.
extern "C" int pk_patch_embed_splitk(const float* video, int B, int C, int F, int H, int W, int ph, int pw, int N, int ngroups,
                                     const void* W0, int ldw0, float* part0, float* stats0, int f00, int nt0, int pt0,
                                     const void* W1, int ldw1, float* part1, float* stats1, int f01, int nt1, int pt1, void* stream) {
    if (!video || B <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || ph <= 0 || pw <= 0 || N <= 0 || N > PW_BN || (ngroups != 1 && ngroups != 2)) return PK_EINVAL;
    if (H % ph || W % pw || (pw & (pw - 1)) || pw < 8 || pw > 128 || (W & 3) || (N & 3)) return PK_EINVAL;
    if (pw < 32 && ph % (32 / pw)) return PK_EINVAL;
    const size_t vbytes = (size_t)B * C * F * H * W * 4;
    if (vbytes >= 0xFFFFFFF0ull) return PK_EINVAL;
    auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
    PatchWideArgs a;
    a.video = video; a.B = B; a.C = C; a.F = F; a.H = H; a.W = W; a.ph = ph; a.pw = pw; a.nh = H / ph; a.nw = W / pw;
    a.N = N; a.video_bytes = (uint32_t)vbytes; a.ngroups = ngroups;
    static const int dbg_env = [] { const char* e = getenv("PK_PATCH_DBG"); return e ? atoi(e) : 0; }();      // timing decomposition only
    a.dbg = dbg_env;
    const void* Ws[2] = {W0, W1}; const int ldws[2] = {ldw0, ldw1}; float* parts[2] = {part0, part1}; float* stt[2] = {stats0, stats1};
    const int f0s[2] = {f00, f01}, nts[2] = {nt0, nt1}, pts[2] = {pt0, pt1};
    int work = 0;
    for (int gi = 0; gi < ngroups; ++gi) {
        PatchWideGroup& G = a.g[gi];
        if (!Ws[gi] || !parts[gi] || !stt[gi] || nts[gi] <= 0 || pts[gi] <= 0 || f0s[gi] < 0 || f0s[gi] + nts[gi] * pts[gi] > F) return PK_EINVAL;
        if (mis(video) || mis(Ws[gi]) || mis(parts[gi]) || (reinterpret_cast<uintptr_t>(stt[gi]) & 7)) return PK_EALIGN;
        G.W = Ws[gi]; G.ldw = ldws[gi]; G.part = parts[gi]; G.stats = stt[gi];
        G.K = C * pts[gi] * ph * pw;
        if (G.K % 64 || G.ldw < G.K || (G.ldw & 7) || (size_t)N * G.ldw * 2 >= 0xFFFFFFF0ull) return PK_EINVAL;
        G.f0 = f0s[gi]; G.nt = nts[gi]; G.pt = pts[gi];
        G.rows = B * nts[gi] * a.nh * a.nw;
        G.mtiles = (G.rows + PE_BM - 1) / PE_BM;
        G.nslices = (G.K + PW_KS - 1) / PW_KS;
        work += G.mtiles * G.nslices;
    }
    if (ngroups == 1) a.g[1] = a.g[0];
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return PK_ELAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_embed_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PW_SMEM) != hipSuccess) return PK_ELAUNCH;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(patch_embed_wide_kernel, dim3(8 * ((work + 7) / 8)), dim3(64 * PE_WAVES), PW_SMEM, reinterpret_cast<hipStream_t>(stream), a);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// tokens of one frame group from its slices: LayerNorm(P) fold (s, t incl. the Linear's bias, eps1; K = P) -> LayerNorm(N) (gamma2, beta2, eps2);
// out2 [.][ldo2] f32 and / or out [.][ldo] bf16; output row = (row / remap_in) * remap_out + remap_off + row % remap_in (remap_in = 0: identity).
extern "C" int pk_patch_embed_finish(const float* part, const float* stats, int nslices, int rows, int N, int K, const float* s, const float* t,
                                     float eps1, const float* gamma2, const float* beta2, float eps2, float* out2, int ldo2, void* out, int ldo,
                                     int remap_in, int remap_out, int remap_off, void* stream) {
    if (!part || !stats || !s || !t || !gamma2 || !beta2 || (!out2 && !out) || nslices <= 0 || rows <= 0 || N <= 0 || N > 512 || (N & 3) || K <= 0) return PK_EINVAL;
    auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
    if (mis(part) || mis(s) || mis(t) || mis(gamma2) || mis(beta2) || (out2 && (mis(out2) || (ldo2 & 3))) || (out && ((reinterpret_cast<uintptr_t>(out) & 7) || (ldo & 3))) ||
        (reinterpret_cast<uintptr_t>(stats) & 7)) return PK_EALIGN;
    if (remap_in < 0 || (remap_in > 0 && (remap_out < remap_in || remap_off < 0))) return PK_EINVAL;
    PatchFinishArgs a{part, stats, nslices, rows, N, K, s, t, eps1, gamma2, beta2, eps2, out2, reinterpret_cast<bf16*>(out), ldo2, ldo, remap_in, remap_out, remap_off};
    hipLaunchKernelGGL(patch_embed_finish_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// the same for 1 or 2 frame groups in ONE launch: g[i] describes group i (the fields of pk_patch_embed_finish), all groups write the same out2 / out
extern "C" int pk_patch_embed_finish_groups(const pk_patch_finish_group* g, int ngroups, int N, float* out2, int ldo2, void* out, int ldo, void* stream) {
    if (!g || (ngroups != 1 && ngroups != 2) || (!out2 && !out) || N <= 0 || N > 512 || (N & 3)) return PK_EINVAL;
    auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
    if ((out2 && (mis(out2) || (ldo2 & 3))) || (out && ((reinterpret_cast<uintptr_t>(out) & 7) || (ldo & 3)))) return PK_EALIGN;
    PatchFinishPair p;
    for (int i = 0; i < ngroups; ++i) {
        const pk_patch_finish_group& d = g[i];
        if (!d.part || !d.stats || !d.s || !d.t || !d.gamma2 || !d.beta2 || d.nslices <= 0 || d.rows <= 0 || d.K <= 0) return PK_EINVAL;
        if (mis(d.part) || mis(d.s) || mis(d.t) || mis(d.gamma2) || mis(d.beta2) || (reinterpret_cast<uintptr_t>(d.stats) & 7)) return PK_EALIGN;
        if (d.remap_in < 0 || (d.remap_in > 0 && (d.remap_out < d.remap_in || d.remap_off < 0))) return PK_EINVAL;
        p.g[i] = PatchFinishArgs{d.part, d.stats, d.nslices, d.rows, N, d.K, d.s, d.t, d.eps1, d.gamma2, d.beta2, d.eps2, out2, reinterpret_cast<bf16*>(out), ldo2, ldo,
                                 d.remap_in, d.remap_out, d.remap_off};
    }
    if (ngroups == 1) p.g[1] = p.g[0];
    p.blocks0 = (p.g[0].rows + 3) / 4;
    const int blocks = p.blocks0 + (ngroups == 2 ? (p.g[1].rows + 3) / 4 : 0);
    hipLaunchKernelGGL(patch_embed_finish_pair_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
