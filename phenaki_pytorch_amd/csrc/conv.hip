// The tokenizer's adversarial branch (reference cvivit.py:59-213 Discriminator, :604-671 losses): the layout kernels that
// turn every nn.Conv2d of the discriminator into a product on the GEMM kernels of this library, plus the small strided-batched product the
// second-order (gradient penalty) graph falls back to.
//
// Images live as CHANNELS-LAST pixel rows  x[(b, y, x)][c]  (f32, C % 4 == 0): a 1x1 convolution is then a plain row GEMM, a k x k one is the
// GEMM of the patch matrix  cols[(b, yo, xo)][(ky, kx, c)]  below, and the reference's  Rearrange('b c (h p1) (w p2) -> b (c p1 p2) h w')  +
// 1x1 convolution (cvivit.py:124-127) is the 2x2 / stride-2 patch matrix against the re-ordered weight.  All of these maps are linear, and
// each comes with its adjoint, so the backward pass AND the backward of the backward pass (the gradient penalty of cvivit.py:59-73
// differentiates the input gradient) stay inside the same five kernels.
#include "common.hpp"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

using pk::f32x4;

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// cols[m][(ky * kw + kx) * C + c] = x[b][yo * stride + ky - pad][xo * stride + kx - pad][c]  (0 outside the image); m = (b, yo, xo)
__global__ void im2col_kernel(const float* __restrict__ x, int H, int W, int C4, int kh, int kw, int stride, int pad, int Ho, int Wo,
                              float* __restrict__ cols, long ldc, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int tap = (int)(r % (kh * kw));
    const long m = r / (kh * kw);
    const int xo = (int)(m % Wo);
    const int yo = (int)((m / Wo) % Ho);
    const long b = m / ((long)Wo * Ho);
    const int ky = tap / kw, kx = tap % kw;
    const int y = yo * stride + ky - pad, xx = xo * stride + kx - pad;
    f32x4 v = zero4();
    if (y >= 0 && y < H && xx >= 0 && xx < W)
        v = *reinterpret_cast<const f32x4*>(x + (((b * H + y) * W + xx) * (long)C4 + c4) * 4);
    *reinterpret_cast<f32x4*>(cols + m * ldc + ((long)tap * C4 + c4) * 4) = v;
}

// the adjoint, in gather form (deterministic: every pixel adds the taps that read it, in tap order):
// dx[b][y][x][c] = sum_{ky, kx : (y + pad - ky) % stride == 0, ...} cols[(b, (y + pad - ky) / stride, (x + pad - kx) / stride)][(ky, kx, c)]
__global__ void col2im_kernel(const float* __restrict__ cols, long ldc, int H, int W, int C4, int kh, int kw, int stride, int pad, int Ho,
                              int Wo, float* __restrict__ dx, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int xx = (int)(r % W);
    const int y = (int)((r / W) % H);
    const long b = r / ((long)W * H);
    f32x4 acc = zero4();
    for (int ky = 0; ky < kh; ++ky) {
        const int ty = y + pad - ky;
        if (ty < 0 || ty % stride) continue;
        const int yo = ty / stride;
        if (yo >= Ho) continue;
        for (int kx = 0; kx < kw; ++kx) {
            const int tx = xx + pad - kx;
            if (tx < 0 || tx % stride) continue;
            const int xo = tx / stride;
            if (xo >= Wo) continue;
            const long m = (b * Ho + yo) * Wo + xo;
            const f32x4 v = *reinterpret_cast<const f32x4*>(cols + m * ldc + ((long)(ky * kw + kx) * C4 + c4) * 4);
            acc += v;
        }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
}

// (B, C, H, W) image -> rows[(b, y, x)][Cp], channels C..Cp-1 zero (the first convolution's K runs over 9 * Cp features: Cp = 8 keeps the
// patch matrix 16-byte aligned for 3-channel frames); and the adjoint (padding channels dropped)
__global__ void nchw_to_rows_kernel(const float* __restrict__ img, int C, long HW, int Cp, float* __restrict__ rows, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per pixel (b, y, x): coalesced plane reads
    if (i >= total) return;
    const long b = i / HW, p = i % HW;
    const float* src = img + b * C * HW + p;
    float* dst = rows + i * Cp;
    for (int c = 0; c < Cp; ++c) dst[c] = c < C ? src[(long)c * HW] : 0.f;
}
__global__ void rows_to_nchw_kernel(const float* __restrict__ rows, int C, long HW, int Cp, float* __restrict__ img, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long b = i / HW, p = i % HW;
    const float* src = rows + i * Cp;
    float* dst = img + b * C * HW + p;
    for (int c = 0; c < C; ++c) dst[(long)c * HW] = src[c];
}

// pick_video_frame (cvivit.py:217-224): img[b][c][:, :] = video[b][c][frame[b]][:, :]; place = its adjoint into a video the caller zeroed
template <bool PLACE>
__global__ void frame_kernel(float* __restrict__ video, const int* __restrict__ frame, int C, int F, long HW4, float* __restrict__ img, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 4 pixels of (b, c)
    if (i >= total) return;
    const long bc = i / HW4, p = i % HW4;
    const long b = bc / C, c = bc % C;
    f32x4* v = reinterpret_cast<f32x4*>(video) + ((b * C + c) * F + frame[b]) * HW4 + p;
    f32x4* g = reinterpret_cast<f32x4*>(img) + i;
    if (PLACE) *v = *g; else *g = *v;
}

// C[z] = op(A[z]) op(B[z]) (+ C[z] if accumulate), exact f32 (one fmaf chain per element, k ascending): 64 x 64 tiles, 16-deep k-steps through
// LDS, 256 threads x (4 x 4) outputs.  Any M, N, K, leading dimension and batch stride; op = transpose by index arithmetic.  This is the
// fallback product of the discriminator's training graph (attention scores per head, single-unit heads, odd shapes) -- the convolutions
// themselves run on pk_gemm.
constexpr int BT = 64, BKK = 16;
__global__ __launch_bounds__(256) void bmm_kernel(const float* __restrict__ A, long lda, long sA, int tA, const float* __restrict__ B, long ldb,
                                                  long sB, int tB, float* __restrict__ Cm, long ldc, long sC, int M, int N, int K, int accumulate) {
    __shared__ float As[BKK][BT + 4];
    __shared__ float Bs[BKK][BT + 4];
    const int z = blockIdx.z;
    A += (long)z * sA; B += (long)z * sB; Cm += (long)z * sC;
    const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += BKK) {
        for (int e = tid; e < BT * BKK; e += 256) {
            // A tile: element (m, k); walk the contiguous index fastest
            int m, k;
            if (tA) { m = e % BT; k = e / BT; } else { k = e % BKK; m = e / BKK; }
            const int gm = m0 + m, gk = k0 + k;
            As[k][m] = (gm < M && gk < K) ? (tA ? A[(long)gk * lda + gm] : A[(long)gm * lda + gk]) : 0.f;
            int n, kb;
            if (tB) { kb = e % BKK; n = e / BKK; } else { n = e % BT; kb = e / BT; }
            const int gn = n0 + n, gkb = k0 + kb;
            Bs[kb][n] = (gn < N && gkb < K) ? (tB ? B[(long)gn * ldb + gkb] : B[(long)gkb * ldb + gn]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BKK; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx * 4 + j;
            if (gn >= N) continue;
            float* c = Cm + (long)gm * ldc + gn;
            *c = accumulate ? *c + acc[i][j] : acc[i][j];
        }
    }
}

inline int nblk(long total) { return (int)((total + 255) / 256); }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int conv_out(int n, int k, int stride, int pad) { return (n + 2 * pad - k) / stride + 1; }

extern "C" int pk_im2col(const float* x, int B, int H, int W, int C, int kh, int kw, int stride, int pad, float* cols, long ldc, void* stream) {
    if (!x || !cols || B <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return PK_EINVAL;
    if ((C & 3) || (ldc & 3) || !al16(x) || !al16(cols)) return PK_EALIGN;
    const int Ho = conv_out(H, kh, stride, pad), Wo = conv_out(W, kw, stride, pad);
    if (Ho <= 0 || Wo <= 0 || ldc < (long)kh * kw * C) return PK_EINVAL;
    const long total = (long)B * Ho * Wo * kh * kw * (C / 4);
    if (total > 0x7FFFFFFFll * 256) return PK_EINVAL;
    hipLaunchKernelGGL(im2col_kernel, dim3(nblk(total)), dim3(256), 0, STREAM(stream), x, H, W, C / 4, kh, kw, stride, pad, Ho, Wo, cols, ldc, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_col2im(const float* cols, long ldc, int B, int H, int W, int C, int kh, int kw, int stride, int pad, float* dx, void* stream) {
    if (!dx || !cols || B <= 0 || H <= 0 || W <= 0 || C <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0) return PK_EINVAL;
    if ((C & 3) || (ldc & 3) || !al16(dx) || !al16(cols)) return PK_EALIGN;
    const int Ho = conv_out(H, kh, stride, pad), Wo = conv_out(W, kw, stride, pad);
    if (Ho <= 0 || Wo <= 0 || ldc < (long)kh * kw * C) return PK_EINVAL;
    const long total = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(col2im_kernel, dim3(nblk(total)), dim3(256), 0, STREAM(stream), cols, ldc, H, W, C / 4, kh, kw, stride, pad, Ho, Wo, dx, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_nchw_to_rows(const float* img, int B, int C, int H, int W, int Cp, float* rows, void* stream) {
    if (!img || !rows || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) return PK_EINVAL;
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(nchw_to_rows_kernel, dim3(nblk(total)), dim3(256), 0, STREAM(stream), img, C, (long)H * W, Cp, rows, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_rows_to_nchw(const float* rows, int B, int C, int H, int W, int Cp, float* img, void* stream) {
    if (!img || !rows || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) return PK_EINVAL;
    const long total = (long)B * H * W;
    hipLaunchKernelGGL(rows_to_nchw_kernel, dim3(nblk(total)), dim3(256), 0, STREAM(stream), rows, C, (long)H * W, Cp, img, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// place = 0: img (B, C, H, W) <- video[:, :, frame[b]];  place = 1: video[:, :, frame[b]] <- img (the caller zeroes the video first)
extern "C" int pk_pick_frames(float* video, const int* frame, int B, int C, int F, int H, int W, float* img, int place, void* stream) {
    if (!video || !frame || !img || B <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0) return PK_EINVAL;
    if (((long)H * W) & 3 || !al16(video) || !al16(img)) return PK_EALIGN;
    const long HW4 = (long)H * W / 4, total = (long)B * C * HW4;
    if (place) hipLaunchKernelGGL(frame_kernel<true>, dim3(nblk(total)), dim3(256), 0, STREAM(stream), video, frame, C, F, HW4, img, total);
    else hipLaunchKernelGGL(frame_kernel<false>, dim3(nblk(total)), dim3(256), 0, STREAM(stream), video, frame, C, F, HW4, img, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_bmm(const float* A, long lda, long sA, int tA, const float* B, long ldb, long sB, int tB, float* C, long ldc, long sC,
                      int batch, int M, int N, int K, int accumulate, void* stream) {
    if (!A || !B || !C || batch <= 0 || batch > 65535 || M <= 0 || N <= 0 || K <= 0) return PK_EINVAL;
    const dim3 grid((N + BT - 1) / BT, (M + BT - 1) / BT, batch);
    if (grid.y > 65535) return PK_EINVAL;
    hipLaunchKernelGGL(bmm_kernel, grid, dim3(256), 0, STREAM(stream), A, lda, sA, tA, B, ldb, sB, tB, C, ldc, sC, M, N, K, accumulate);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
