// pk_gemm: every nn.Linear on the hot path (reference attention.py:50-52,117-119, cvivit.py:276,283,
// 328,333, phenaki_pytorch.py:147) as one MFMA GEMM with a fused epilogue:
//   C = act(A @ W^T + bias) (+ residual)        act in {none, GEGLU pair, LeakyReLU(0.1)}
// Roofline: MFMA-bound (bf16 2.5 PFLOP/s dense, exact-f32 157 TFLOP/s); algorithmic flops 2*M*N*K.
#include "gemm_core.hpp"

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
};

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

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, g = lane >> 4, lr = lane & 15;
    float* Cf = reinterpret_cast<float*>(e.C);
    T* Ct = reinterpret_cast<T*>(e.C);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 16 * TM + i * 16 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * 16 * TN + j * 16 + g * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[i][j];
            if (e.vec_ok) {
                if (e.bias) v += *reinterpret_cast<const f32x4*>(e.bias + n);
                if (e.act == ACT_GEGLU) {
                    const float o0 = gelu_erf(v[1]) * v[0], o1 = gelu_erf(v[3]) * v[2];
                    const size_t o = (size_t)m * e.ldc + (n >> 1);
                    if (e.out_f32) store2(Cf + o, o0, o1); else store2(Ct + o, o0, o1);
                } else {
                    if (e.act == ACT_LEAKY) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.1f * v[r];
                    }
                    if (e.res) v += *reinterpret_cast<const f32x4*>(e.res + (size_t)m * e.ldr + n);
                    const size_t o = (size_t)m * e.ldc + n;
                    if (e.out_f32) store4(Cf + o, v); else store4(Ct + o, v);
                }
            } else {
                // scalar path (N not a multiple of 4, e.g. heads = 2 or a 1-wide critic head)
                for (int r = 0; r < 4; ++r) {
                    const int nn = n + r;
                    if (nn >= p.N) break;
                    float x = v[r] + (e.bias ? e.bias[nn] : 0.f);
                    if (e.act == ACT_GEGLU) {
                        if (r & 1) continue;
                        const float gate = (nn + 1 < p.N) ? v[r + 1] + (e.bias ? e.bias[nn + 1] : 0.f) : 0.f;
                        x = gelu_erf(gate) * x;
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
}

template <typename T, typename TA>
static int launch_gemm(const GemmOperands& p, const GemmEpilogue& e, hipStream_t s) {
    const long blocks128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (blocks128 >= 384) {
        using Tile = GemmTile<T, TA, 4, 4>;
        dim3 grid((p.M + Tile::BM - 1) / Tile::BM, (p.N + Tile::BN - 1) / Tile::BN);
        hipLaunchKernelGGL((gemm_kernel<T, TA, 4, 4>), grid, dim3(256), Tile::SMEM, s, p, e);
    } else {
        using Tile = GemmTile<T, TA, 2, 2>;
        dim3 grid((p.M + Tile::BM - 1) / Tile::BM, (p.N + Tile::BN - 1) / Tile::BN);
        hipLaunchKernelGGL((gemm_kernel<T, TA, 2, 2>), grid, dim3(256), Tile::SMEM, s, p, e);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}

}  // namespace pk

using namespace pk;

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int pk_gemm(int dtype, int a_is_f32, const void* A, int lda, const void* W, int ldw,
                       int M, int N, int K, const float* bias, const float* res, int ldr,
                       void* C, int ldc, int out_is_f32, int act, const int* a_rows, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !C) return PK_EINVAL;
    if (dtype != 0 && dtype != 1) return PK_EINVAL;
    if (act < 0 || act > 2) return PK_EINVAL;
    const int eps_w = dtype == 1 ? 8 : 4;                        // elements per 16 B of W
    const int eps_a = (dtype == 1 && !a_is_f32) ? 8 : 4;         // alignment quantum of A rows
    if (K % eps_w || ldw % eps_w || lda % eps_a) return PK_EALIGN;
    if (dtype == 1 && a_is_f32 && (K % 8)) return PK_EALIGN;
    if (!al16(A) || !al16(W)) return PK_EALIGN;
    if (dtype == 0 && !a_is_f32) return PK_EINVAL;               // exact-f32 mode keeps everything f32
    if (act == ACT_GEGLU && (N & 1)) return PK_EINVAL;
    GemmOperands p{A, W, a_rows, lda, ldw, M, N, K};
    GemmEpilogue e{bias, res, C, ldr, ldc, out_is_f32, act, 0};
    const int out_el = out_is_f32 ? 4 : (dtype == 1 ? 2 : 4);
    bool v = (N % 4 == 0) && (ldc % 4 == 0) && al16(C) && (!bias || al16(bias)) && (!res || (al16(res) && ldr % 4 == 0));
    if (act == ACT_GEGLU) v = v && (ldc % 2 == 0) && ((reinterpret_cast<uintptr_t>(C) & 7) == 0);
    (void)out_el;
    e.vec_ok = v ? 1 : 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == 1) return a_is_f32 ? launch_gemm<bf16, float>(p, e, s) : launch_gemm<bf16, bf16>(p, e, s);
    return launch_gemm<float, float>(p, e, s);
}
