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
    const int f = frame[b];
    f32x4* g = reinterpret_cast<f32x4*>(img) + i;
    if ((unsigned)f >= (unsigned)F) {                                    // an index outside [0, F) never becomes an out-of-bounds access:
        if (!PLACE) *g = f32x4{0.f, 0.f, 0.f, 0.f};                      // the picked frame reads as zeros, the adjoint drops the write
        return;
    }
    f32x4* v = reinterpret_cast<f32x4*>(video) + ((b * C + c) * F + f) * HW4 + p;
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

// ---- row-wise pieces of the discriminator's attention block WITH their second derivatives (the gradient penalty differentiates the input gradient through
// the block, cvivit.py:59-73, 166-168): softmax over the keys, l2norm x scale vector (attention.py:153-155), and the backward-of-backward of the gamma-only
// LayerNorm (attention.py:29-36; its forward / backward are pk_layernorm / pk_layernorm_bwd).  One wave per row, lanes stride the row; the sums over ROWS that
// the parameter gradients need (d scale, d gamma) are left as per-row contributions for pk_colsum (deterministic two-stage sum).  Formulas (u = upstream):
//   softmax      y = softmax(x);  B(y, dy) = y (dy - <y, dy>);  d<u, B>/dy_in = u (dy - <y, dy>) - dy <u, y>,  d<u, B>/d dy = B(y, u)
//   l2 x scale   z = x^ sc, x^ = x / |x|;  dx = P t / |x| (t = dz sc, P = I - x^ x^T), dsc = sum_rows dz x^;
//                second order: grad_dz = sc P gx / |x| + gsc x^,  grad_sc = sum_rows dz P gx / |x|,
//                              grad_x = -[P (a gx + b t) + x^ <gx, t - x^ a>] / |x|^2 + P (gsc dz) / |x|,  a = <x^, t>, b = <gx, x^>
//   LayerNorm    dx = (g^ - mean g^ - x^ mean(g^ x^)) / sigma (g^ = dy gamma), dgamma = sum_rows dy x^;
//                second order (u for dx, w for dgamma): gg = (u - mean u - x^ mean(u x^)) / sigma, grad_dy = gamma gg + w x^, grad_gamma = sum_rows dy gg,
//                              v = -(m2 u + c2 g^) / sigma + w dy,  grad_x = (v - mean v - x^ mean(v x^)) / sigma - S x^ / (D sigma^2),
//                              S = <u, g^> - D c1 m1 - D c2 m2,  c1 = mean u, c2 = mean(u x^), m1 = mean g^, m2 = mean(g^ x^)
// (each checked against torch.autograd on the CPU to 1e-15 before it was written down here, and against torch on the GPU in tests/test_gan_gpu.py)
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#define PK_ROW_PROLOGUE \
    const int lane = threadIdx.x & 63; \
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); \
    if (row >= M) return;

// mode 0: y = softmax(a);  1: out = a (b - <a, b>)  [a = y, b = dy];  2: out = c (b - <a, b>) - b <c, a>  [a = y, b = dy, c = upstream]
__global__ __launch_bounds__(256) void row_softmax_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                          float* __restrict__ out, long M, int n, int mode) {
    PK_ROW_PROLOGUE
    const float* ar = a + row * n;
    float* orow = out + row * n;
    if (mode == 0) {
        float mx = -INFINITY;
        for (int j = lane; j < n; j += 64) mx = fmaxf(mx, ar[j]);
        mx = wmax(mx);
        float sm = 0.f;
        for (int j = lane; j < n; j += 64) sm += __expf(ar[j] - mx);
        sm = wsum(sm);
        const float inv = 1.0f / sm;
        for (int j = lane; j < n; j += 64) orow[j] = __expf(ar[j] - mx) * inv;
        return;
    }
    const float* br = b + row * n;
    float r = 0.f, q = 0.f;
    for (int j = lane; j < n; j += 64) { r += ar[j] * br[j]; if (mode == 2) q += c[row * n + j] * ar[j]; }
    r = wsum(r);
    if (mode == 2) q = wsum(q);
    for (int j = lane; j < n; j += 64)
        orow[j] = mode == 1 ? ar[j] * (br[j] - r) : c[row * n + j] * (br[j] - r) - br[j] * q;
}

// mode 0: z = x^ sc;  1: dx, dsc_rows from dz;  2: grad_x, grad_sc_rows, grad_dz from (dz, gx, gsc)
__global__ __launch_bounds__(256) void row_l2scale_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ dz,
                                                          const float* __restrict__ gx, const float* __restrict__ gsc, float* __restrict__ o0,
                                                          float* __restrict__ o1, float* __restrict__ o2, long M, int d, int mode) {
    PK_ROW_PROLOGUE
    const float* xr = x + row * d;
    float ss = 0.f;
    for (int j = lane; j < d; j += 64) ss += xr[j] * xr[j];
    const float nrm = fmaxf(sqrtf(wsum(ss)), 1e-12f), inv = 1.0f / nrm;      // F.normalize eps
    if (mode == 0) {
        for (int j = lane; j < d; j += 64) o0[row * d + j] = xr[j] * inv * sc[j];
        return;
    }
    const float* dzr = dz + row * d;
    float a = 0.f;                                                        // <x^, t>, t = dz sc
    for (int j = lane; j < d; j += 64) a += xr[j] * inv * dzr[j] * sc[j];
    a = wsum(a);
    if (mode == 1) {
        for (int j = lane; j < d; j += 64) {
            const float u = xr[j] * inv, t = dzr[j] * sc[j];
            o0[row * d + j] = (t - u * a) * inv;
            o1[row * d + j] = dzr[j] * u;
        }
        return;
    }
    const float* gr = gx + row * d;
    float b = 0.f, gt = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
    // b = <gx, x^>; gt = <gx, t>; e1 = <x^, a gx + b t> needs b first: two passes
    for (int j = lane; j < d; j += 64) { const float u = xr[j] * inv; b += gr[j] * u; gt += gr[j] * dzr[j] * sc[j]; e3 += u * gsc[j] * dzr[j]; }
    b = wsum(b); gt = wsum(gt); e3 = wsum(e3);
    e1 = a * b + b * a;                                                   // <x^, a gx + b t> = a b + b a
    e2 = gt - b * a;                                                      // <gx, t - x^ a>
    for (int j = lane; j < d; j += 64) {
        const float u = xr[j] * inv, t = dzr[j] * sc[j];
        const float pg = gr[j] - u * b;                                   // (P gx)_j
        o2[row * d + j] = sc[j] * pg * inv + gsc[j] * u;                  // grad_dz
        o1[row * d + j] = dzr[j] * pg * inv;                              // grad_sc contribution of this row
        const float w_ = a * gr[j] + b * t;
        const float pw = w_ - u * e1;
        const float pgd = gsc[j] * dzr[j] - u * e3;                       // (P (gsc dz))_j
        o0[row * d + j] = -(pw + u * e2) * inv * inv + pgd * inv;         // grad_x
    }
}

// backward-of-backward of the gamma-only LayerNorm: (grad_x, grad_gamma_rows, grad_dy) from (x, gamma, dy, u, w)
__global__ __launch_bounds__(256) void row_ln_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                          const float* __restrict__ u, const float* __restrict__ w, float eps, float* __restrict__ gxo,
                                                          float* __restrict__ ggo, float* __restrict__ gdyo, long M, int D) {
    PK_ROW_PROLOGUE
    const float* xr = x + row * D;
    const float* dyr = dy + row * D;
    const float* ur = u + row * D;
    const float invD = 1.0f / (float)D;
    float s1 = 0.f;
    for (int j = lane; j < D; j += 64) s1 += xr[j];
    const float mu = wsum(s1) * invD;
    float s2 = 0.f;
    for (int j = lane; j < D; j += 64) { const float c = xr[j] - mu; s2 += c * c; }
    const float sig = sqrtf(wsum(s2) * invD + eps), isig = 1.0f / sig;
    float c1 = 0.f, c2 = 0.f, m1 = 0.f, m2 = 0.f, ug = 0.f;
    for (int j = lane; j < D; j += 64) {
        const float xh = (xr[j] - mu) * isig, gh = dyr[j] * gamma[j];
        c1 += ur[j]; c2 += ur[j] * xh; m1 += gh; m2 += gh * xh; ug += ur[j] * gh;
    }
    c1 = wsum(c1) * invD; c2 = wsum(c2) * invD; m1 = wsum(m1) * invD; m2 = wsum(m2) * invD; ug = wsum(ug);
    const float S = ug - (float)D * c1 * m1 - (float)D * c2 * m2;
    float v1 = 0.f, v2 = 0.f;
    for (int j = lane; j < D; j += 64) {
        const float xh = (xr[j] - mu) * isig, gh = dyr[j] * gamma[j];
        const float v = -(m2 * ur[j] + c2 * gh) * isig + w[j] * dyr[j];
        v1 += v; v2 += v * xh;
    }
    v1 = wsum(v1) * invD; v2 = wsum(v2) * invD;
    for (int j = lane; j < D; j += 64) {
        const float xh = (xr[j] - mu) * isig, gh = dyr[j] * gamma[j];
        const float gg = (ur[j] - c1 - xh * c2) * isig;
        gdyo[row * D + j] = gamma[j] * gg + w[j] * xh;
        ggo[row * D + j] = dyr[j] * gg;
        const float v = -(m2 * ur[j] + c2 * gh) * isig + w[j] * dyr[j];
        gxo[row * D + j] = (v - v1 - xh * v2) * isig - S * isig * isig * xh * invD;
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

extern "C" int pk_row_softmax(const float* a, const float* b, const float* c, float* out, long M, int n, int mode, void* stream) {
    if (!a || !out || M <= 0 || n <= 0 || mode < 0 || mode > 2 || (mode >= 1 && !b) || (mode == 2 && !c)) return PK_EINVAL;
    hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, STREAM(stream), a, b, c, out, M, n, mode);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_row_l2scale(const float* x, const float* sc, const float* dz, const float* gx, const float* gsc, float* o0, float* o1, float* o2,
                              long M, int d, int mode, void* stream) {
    if (!x || !sc || !o0 || M <= 0 || d <= 0 || mode < 0 || mode > 2) return PK_EINVAL;
    if ((mode >= 1 && (!dz || !o1)) || (mode == 2 && (!gx || !gsc || !o2))) return PK_EINVAL;
    hipLaunchKernelGGL(row_l2scale_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, STREAM(stream), x, sc, dz, gx, gsc, o0, o1, o2, M, d, mode);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_row_ln_bwd2(const float* x, const float* gamma, const float* dy, const float* u, const float* w, float eps, float* grad_x,
                              float* grad_gamma_rows, float* grad_dy, long M, int D, void* stream) {
    if (!x || !gamma || !dy || !u || !w || !grad_x || !grad_gamma_rows || !grad_dy || M <= 0 || D <= 0) return PK_EINVAL;
    hipLaunchKernelGGL(row_ln_bwd2_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, STREAM(stream), x, gamma, dy, u, w, eps, grad_x, grad_gamma_rows,
                       grad_dy, M, D);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
