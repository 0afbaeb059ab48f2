// pk_qkv_project (bf16 / split-bf16): to_q and to_kv (reference attention.py:142-146) as ONE MFMA GEMM launch whose epilogue is the whole
// attention pre-processing of attention.py:146-157 -- head split, l2norm of every 64-wide head row, q_scale / k_scale,
// the similarity scale folded into q, V stored transposed -- writing the operand images pk_attn_fwd consumes
//   Qp [S][h][nq_pad][64]   Kp [S][h][nk_pad][64]   Vt [S][h][64][nk_pad]     (bf16; split-bf16: the pre-split bf16x3p images, common.hpp)
// so the f32 q / kv matrices never exist in HBM and pk_attn_prep's two launches disappear.
// Tile: 64 rows x 64 columns = ONE head, 4 waves stacked along the rows (wave tile 16 x 64): a wave holds whole head rows,
// the l2 norm is a 16-value in-lane sum plus a 4-lane-group shuffle.  Main loop: gemm_dma.hpp (LDS-DMA ring of 2).
// Column tiles [0, h) are q heads (A = LayerNorm(x)), [h, 2h) k heads and [2h, 3h) v heads (A = the un-normalised x).
#include "gemm_dma.hpp"

namespace pk {

struct QkvArgs {
    const void* xq; const void* xkv;        // [M][ld] bf16; xkv may be null (query side only)
    const void* wq; const void* wkv;        // [h*64][ldw], [2*h*64][ldw] bf16, K zero-padded to the k-tile
    int ld, ldw;
    int M, K, h, nseq;                      // M = S * nseq rows
    const float* q_scale; const float* k_scale; float scale;
    void* Qp; void* Kp; void* Vt;
    int nq_pad, nk_pad;
    int krot;
    const float* q_ln_s;                    // LayerNorm folded into to_q: xq holds the UN-normalised rows, wq = gamma (.) Wq, q_ln_s[n] = sum_k wq[n][k]
};

// TM = 1: 64-row tiles (wave tile 16 x 64); TM = 2: 128-row tiles (wave tile 32 x 64) -- the kernel is bound by the L1 -> LDS fill path
// (3456 tiles x 128 KB = 442 MB per launch at 2 x 8 x 576 rows) and a 128-row tile moves 24 KB per k-tile for the work of two 64-row
// tiles (2 x 16 KB): a quarter less fill traffic, still 48 KB of LDS = 3 workgroups per CU
// T = bf16x3 (round 4): x rows are f32 and split in registers, the weights are host-split planes (gemm_core.hpp), the images are written pre-split
// (store4 / store_elem of bf16x3p) -- the exact-f32 LayerNorm + to_q GEMM + to_kv GEMM + pk_attn_prep of that mode become this one launch.
template <typename T, int TM> using QkvTileT = GemmDma<T, TM, 4, 4, 1, 2, 128>;

template <typename T, int TM>
__global__ __launch_bounds__(256) void qkv_project_kernel(const QkvArgs a) {
    using QkvTile = QkvTileT<T, TM>;
    typedef typename ImageOf<T>::type TI;
    constexpr int BM = 64 * TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware tile map (see gemm.hip): XCD x owns a contiguous chunk of row tiles and walks all column tiles for it
    const int MT = (a.M + BM - 1) / BM, cmax = (MT + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mstart = xcd * MT / 8, mcount = (xcd + 1) * MT / 8 - mstart;
    const int ml = idx % cmax;
    if (ml >= mcount) return;
    const int m0 = (mstart + ml) * BM;
    const int nt = idx / cmax;                                  // 0..3h-1
    const bool is_q = nt < a.h;
    const int kind = is_q ? 0 : (nt < 2 * a.h ? 1 : 2);         // 0 q, 1 k, 2 v
    const int hh = nt - kind * a.h;
    GemmOperands p;
    p.A = is_q ? a.xq : a.xkv;
    p.W = is_q ? a.wq : a.wkv;
    p.a_rows = nullptr;
    p.lda = a.ld; p.ldw = a.ldw;
    p.M = a.M; p.N = is_q ? a.h * 64 : 2 * a.h * 64; p.K = a.K;
    p.plain_map = 0;
    p.krot = a.krot;
    p.w_gap_from = 0; p.w_gap_rows = 0; p.panel = 0;
    const int n0 = is_q ? hh * 64 : (kind == 1 ? hh * 64 : (a.h + hh) * 64);

    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, lr = lane & 15;
    if (is_q && a.q_ln_s) {
        // q = l2norm(LN(x) Wq^T): LN(x) Wq^T = rstd * (x (gamma.Wq)^T - mean * s) and the l2norm cancels rstd > 0, so only the row
        // mean (from the A fragments of the main loop) and s are needed
        float rsum[TM], rsq[TM];
        (void)QkvTile::template run_stats<1>(p, a.M, m0, n0, smem, acc, rsum, rsq);
        f32x4 s4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) s4[j] = *reinterpret_cast<const f32x4*>(a.q_ln_s + n0 + j * 16 + g * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float mean = rsum[i] / (float)a.K;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] -= mean * s4[j][r];
        }
    } else {
        (void)QkvTile::run(p, a.M, m0, n0, smem, acc);
    }

    const float* sc = kind == 0 ? a.q_scale : a.k_scale;
    f32x4 scv[4];                                               // all loads before the first store (in-order vmcnt)
    if (kind != 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) scv[j] = *reinterpret_cast<const f32x4*>(sc + j * 16 + g * 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wave * 16 * TM + i * 16 + lr;
        if (m >= a.M) continue;
        const int s = m / a.nseq, pos = m % a.nseq;
        const size_t sh = (size_t)s * a.h + hh;
        if (kind == 2) {
            // V^T: element (key = pos, d) -> Vt[sh][d][pos]
            TI* vt = reinterpret_cast<TI*>(a.Vt) + sh * 64 * a.nk_pad + pos;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) store_elem(vt + (size_t)(j * 16 + g * 4 + r) * a.nk_pad, acc[i][j][r]);
            continue;
        }
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ss += acc[i][j][r] * acc[i][j][r];
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float inv = (kind == 0 ? a.scale : 1.0f) / fmaxf(sqrtf(ss), 1e-12f);        // F.normalize eps = 1e-12
        TI* dst = kind == 0 ? reinterpret_cast<TI*>(a.Qp) + (sh * a.nq_pad + pos) * 64
                            : reinterpret_cast<TI*>(a.Kp) + (sh * a.nk_pad + pos) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = acc[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= inv * scv[j][r];
            store4(dst + j * 16 + g * 4, v);
        }
    }
}

}  // namespace pk
using namespace pk;

// dtype 1: bf16 (xq / xkv / weights bf16, K-tile 64); dtype 2: split-bf16 (xq / xkv f32 rows, weights = host-split planes counted in 4-byte
// units, K-tile 32, images written as bf16x3p).  xq [M][ld] = LayerNorm(x) rows, xkv [M][ld] = x rows (or NULL: query side only); M = S * nseq.
// q_ln_s != NULL: the LayerNorm is folded into to_q -- xq holds the un-normalised rows, wq = gamma (.) Wq, q_ln_s its row sums.
template <typename T>
static int qkv_project_launch(const QkvArgs& a, int NT, hipStream_t st) {
    // 128-row tiles once there are >= 4 of them per CU (tuning knob PK_QKV_TM: 1 / 2).  tools/qkv_bench.py, n = 576, 8 heads, K = 512:
    // S = 16 (1728 tiles of 128 rows) 29.4 vs 35.2 us; S = 8 (864) 18.4 vs 17.7 us; S = 4: 11.5 vs 11.2 us
    static const int tm_env = [] { const char* e = getenv("PK_QKV_TM"); return e ? atoi(e) : 0; }();
    const long M = a.M;
    const long tiles128 = ((M + 127) / 128) * NT;
    const bool big = tm_env ? tm_env == 2 : tiles128 >= 4 * 256;
    if (big) {
        const int MT = (int)((M + 127) / 128);
        constexpr int lds = QkvTileT<T, 2>::SMEM;
        hipLaunchKernelGGL((qkv_project_kernel<T, 2>), dim3(8 * ((MT + 7) / 8) * NT), dim3(256), lds, st, a);
    } else {
        const int MT = (int)((M + 63) / 64);
        constexpr int lds = QkvTileT<T, 1>::SMEM;
        hipLaunchKernelGGL((qkv_project_kernel<T, 1>), dim3(8 * ((MT + 7) / 8) * NT), dim3(256), lds, st, a);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_qkv_project(int dtype, const void* xq, const void* xkv, int ld, const void* wq, const void* wkv, int ldw,
                              int S, int nseq, int h, int K, const float* q_scale, const float* k_scale, float scale,
                              void* Qp, void* Kp, void* Vt, int nq_pad, int nk_pad, const float* q_ln_s, void* stream) {
    if (dtype != 1 && dtype != 2) return PK_EINVAL;
    if (!xq || !wq || !q_scale || !Qp || S <= 0 || nseq <= 0 || h <= 0 || K <= 0) return PK_EINVAL;
    if (xkv && (!wkv || !k_scale || !Kp || !Vt)) return PK_EINVAL;
    if (nq_pad < nseq || (xkv && nk_pad < nseq)) return PK_EINVAL;
    auto mis = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) != 0; };
    const int q = dtype == 1 ? 7 : 3, bk = dtype == 1 ? 64 : 32, esz = dtype == 1 ? 2 : 4;
    if ((K & q) || (ld & q) || (ldw & q) || mis(xq) || mis(wq) || mis(q_scale) || mis(Qp) ||
        (xkv && (mis(xkv) || mis(wkv) || mis(k_scale) || mis(Kp) || (nk_pad & 3)))) return PK_EALIGN;
    if (dtype == 2 && ((nk_pad & 31) || (reinterpret_cast<uintptr_t>(Qp) & 127) || (xkv && ((reinterpret_cast<uintptr_t>(Kp) | reinterpret_cast<uintptr_t>(Vt)) & 127))))
        return PK_EALIGN;                                       // pre-split images: 128-byte blocks of 32 elements
    if (ldw < (K + bk - 1) / bk * bk) return PK_EINVAL;         // W zero-padded along K to the k-tile
    const long M = (long)S * nseq;
    if ((size_t)M * ld * esz >= 0xFFFFFFF0ull) return PK_EINVAL;
    if (q_ln_s && mis(q_ln_s)) return PK_EALIGN;
    QkvArgs a{xq, xkv, wq, wkv, ld, ldw, (int)M, K, h, nseq, q_scale, k_scale, scale, Qp, Kp, Vt, nq_pad, nk_pad, 0, q_ln_s};
    const int NT = xkv ? 3 * h : h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return dtype == 1 ? qkv_project_launch<bf16>(a, NT, st) : qkv_project_launch<bf16x3>(a, NT, st);
}
