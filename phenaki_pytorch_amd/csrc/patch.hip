// 3-D patch / un-patch of (B, C, F, H, W) f32 video (reference cvivit.py:273-285 Rearrange +
// nn.LayerNorm(P), cvivit.py:326-334 Rearrange back).  Feature order inside a patch vector is
// (c, pt, p1, p2): runs of p2 consecutive floats along W.  HBM-bound: the video is read (written)
// exactly once with 16-byte accesses; algorithmic bytes per patch row = 4*P in + elt*P out.
#include "common.hpp"

namespace pk {

struct PatchGeom {
    int B, C, F, H, W;     // video dims
    int f0, nt;            // first frame of this group, number of temporal patches
    int pt, ph, pw;        // patch extents
    int nh, nw;            // patches per frame (H/ph, W/pw)
};

__device__ __forceinline__ size_t patch_elem_offset(const PatchGeom& g, int b, int tt, int hh, int ww, int k) {
    // k = ((c*pt + dt)*ph + y)*pw + x
    const int x = k % g.pw; int r = k / g.pw;
    const int y = r % g.ph; r /= g.ph;
    const int dt = r % g.pt; const int c = r / g.pt;
    return ((((size_t)b * g.C + c) * g.F + (g.f0 + tt * g.pt + dt)) * g.H + (hh * g.ph + y)) * g.W + (ww * g.pw + x);
}

// one 256-thread block per patch row; P/4 <= 256*VMAX.  The (c, dt, y) -> video offset of every patch row is put in an
// LDS table once per block (one integer-division chain per ROW, not per 16-byte piece: with the divisions in the load
// loop the kernel needed 198 VGPRs -> 2 waves/SIMD and ran at 1.9 TB/s); all loads are issued before the first use.
constexpr int PATCH_MAX_ROWS = 2048;

template <typename TO, int VMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void patchify_ln_kernel(const float* __restrict__ video, PatchGeom g,
                                                          const float* __restrict__ weight, const float* __restrict__ bias,
                                                          float eps, TO* __restrict__ out, int ldo) {
    __shared__ float red[8];
    __shared__ uint32_t roff[PATCH_MAX_ROWS];
    const int row = blockIdx.x;                      // ((b*nt + tt)*nh + hh)*nw + ww
    int r = row;
    const int ww = r % g.nw; r /= g.nw;
    const int hh = r % g.nh; r /= g.nh;
    const int tt = r % g.nt; const int b = r / g.nt;
    const int R = g.C * g.pt * g.ph, P = R * g.pw, nv = P >> 2, pw4 = g.pw >> 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int pr = threadIdx.x; pr < R; pr += 256) {
        const int y = pr % g.ph, cd = pr / g.ph;
        const int dt = cd % g.pt, c = cd / g.pt;
        roff[pr] = ((uint32_t)(c * g.F + dt) * g.H + y) * g.W;
    }
    const float* base = video + ((((size_t)b * g.C) * g.F + (g.f0 + tt * g.pt)) * g.H + hh * g.ph) * g.W + ww * g.pw;
    __syncthreads();
    f32x4 v[VMAX];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nv) {
            const int pr = c / pw4;
            v[i] = *reinterpret_cast<const f32x4*>(base + roff[pr] + (c - pr * pw4) * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < VMAX; ++i)
        if (threadIdx.x + i * 256 < nv) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)P;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)P + eps);
#pragma unroll
    for (int i = 0; i < VMAX; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nv) {
            if (!weight) {                                    // no LayerNorm: the raw patch rows (the training step keeps them, train_cvivit.py)
                store4(out + (size_t)row * ldo + c * 4, v[i]);
                continue;
            }
            const f32x4 wv = *reinterpret_cast<const f32x4*>(weight + c * 4);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c * 4);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
            store4(out + (size_t)row * ldo + c * 4, y);
        }
    }
}

// pix [rows][P] f32 (+ optional nothing else) -> video; one thread per 4 consecutive x
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ pix, int ldp, PatchGeom g,
                                                         float* __restrict__ video, long total_vec) {
    const int P = g.C * g.pt * g.ph * g.pw, nv = P >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total_vec; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % nv);
        int r = (int)(idx / nv);
        const int row = r;
        const int ww = r % g.nw; r /= g.nw;
        const int hh = r % g.nh; r /= g.nh;
        const int tt = r % g.nt; const int b = r / g.nt;
        const f32x4 v = *reinterpret_cast<const f32x4*>(pix + (size_t)row * ldp + c * 4);
        *reinterpret_cast<f32x4*>(video + patch_elem_offset(g, b, tt, hh, ww, c * 4)) = v;
    }
}


// dst[row][col] = fmask[b][f0 + tt*pt + dt] ? src[row][col] : 0 on a (rows, P) patch-layout matrix (row = (b, tt, hh, ww), col = (c, dt, y, x)):
// the frame mask of variable-length training (cvivit.py:585-589) applied where the reconstruction loss is taken in patch layout
__global__ __launch_bounds__(256) void patch_frame_mask_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, PatchGeom g,
                                                               const unsigned char* __restrict__ fmask, long total_vec) {
    const int P = g.C * g.pt * g.ph * g.pw, nv = P >> 2, per_dt = g.ph * g.pw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total_vec; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % nv);
        const long row = idx / nv;
        const long bt = row / ((long)g.nh * g.nw);
        const int tt = (int)(bt % g.nt), b = (int)(bt / g.nt);
        const int dt = ((c4 * 4) / per_dt) % g.pt;
        const bool keep = fmask[(long)b * g.F + g.f0 + tt * g.pt + dt] != 0;
        const f32x4 v = keep ? *reinterpret_cast<const f32x4*>(src + row * lds + c4 * 4) : f32x4{0, 0, 0, 0};
        *reinterpret_cast<f32x4*>(dst + row * ldd + c4 * 4) = v;
    }
}

// sum over kept frames of (a - b)^2 for two (B, C, F, H, W) f32 videos (cvivit.py:585-591 F.mse_loss numerator): every
// workgroup folds a grid-stride slice into ONE double partial (deterministic: fixed slice per workgroup, fixed tree), the
// caller adds the <= 1024 partials.  fmask (B*F bytes, 1 = keep) or null.
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const unsigned char* __restrict__ fmask, int C, int F, long hw4,
                                                             long total4, double* __restrict__ partials) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        if (fmask) {
            const long plane = idx / hw4;                      // (b*C + c)*F + f
            const int f = (int)(plane % F);
            const long bb = plane / ((long)C * F);
            if (!fmask[bb * F + f]) continue;
        }
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + idx * 4), y = *reinterpret_cast<const f32x4*>(b + idx * 4);
        const f32x4 d = x - y;
        acc += (double)((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace pk
using namespace pk;

static int check_geom(const PatchGeom& g) {
    if (g.B <= 0 || g.C <= 0 || g.F <= 0 || g.nt <= 0 || g.pt <= 0 || g.ph <= 0 || g.pw <= 0) return PK_EINVAL;
    if (g.H % g.ph || g.W % g.pw || g.f0 < 0 || g.f0 + g.nt * g.pt > g.F) return PK_EINVAL;
    if ((g.pw & 3) || (g.W & 3)) return PK_EALIGN;
    return PK_OK;
}

// video (B,C,F,H,W) f32, frames [f0, f0 + nt*pt) -> out[(b,tt,hh,ww)][P] = LayerNorm_P(patch) (f32 or bf16); weight = bias = null: the raw patch
extern "C" int pk_patchify_ln(const float* video, int B, int C, int F, int H, int W, int f0, int nt,
                              int pt, int ph, int pw, const float* weight, const float* bias, float eps,
                              void* out, int ldo, int out_kind, void* stream) {
    PatchGeom g{B, C, F, H, W, f0, nt, pt, ph, pw, ph ? H / ph : 0, pw ? W / pw : 0};
    if (int rc = check_geom(g)) return rc;
    if (!video || (!weight != !bias) || !out || (ldo & 3)) return PK_EINVAL;
    const int P = C * pt * ph * pw;
    if ((P >> 2) > 256 * 8 || C * pt * ph > PATCH_MAX_ROWS || (size_t)C * F * H * W >= 0x7FFFFFFFull) return PK_EINVAL;
    const int rows = B * nt * g.nh * g.nw;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool small = (P >> 2) <= 256 * 4;
    if (out_kind == 0) {
        if (small) hipLaunchKernelGGL((patchify_ln_kernel<float, 4>), dim3(rows), dim3(256), 0, s, video, g, weight, bias, eps, (float*)out, ldo);
        else hipLaunchKernelGGL((patchify_ln_kernel<float, 8>), dim3(rows), dim3(256), 0, s, video, g, weight, bias, eps, (float*)out, ldo);
    } else {
        if (small) hipLaunchKernelGGL((patchify_ln_kernel<bf16, 4>), dim3(rows), dim3(256), 0, s, video, g, weight, bias, eps, (bf16*)out, ldo);
        else hipLaunchKernelGGL((patchify_ln_kernel<bf16, 8>), dim3(rows), dim3(256), 0, s, video, g, weight, bias, eps, (bf16*)out, ldo);
    }
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_patch_frame_mask(const float* src, long lds, float* dst, long ldd, const unsigned char* fmask, int B, int C, int F, int H, int W,
                                   int f0, int nt, int pt, int ph, int pw, void* stream) {
    PatchGeom g{B, C, F, H, W, f0, nt, pt, ph, pw, ph ? H / ph : 0, pw ? W / pw : 0};
    if (int rc = check_geom(g)) return rc;
    if (!src || !dst || !fmask || (lds & 3) || (ldd & 3) || ((ph * pw) & 3)) return PK_EINVAL;
    const int P = C * pt * ph * pw;
    const long total = (long)B * nt * g.nh * g.nw * (P >> 2);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(patch_frame_mask_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, lds, dst, ldd, g, fmask, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// pix[(b,tt,hh,ww)][P] f32 -> video (B,C,F,H,W) frames [f0, f0 + nt*pt)
extern "C" int pk_unpatchify(const float* pix, int ldp, float* video, int B, int C, int F, int H, int W,
                             int f0, int nt, int pt, int ph, int pw, void* stream) {
    PatchGeom g{B, C, F, H, W, f0, nt, pt, ph, pw, ph ? H / ph : 0, pw ? W / pw : 0};
    if (int rc = check_geom(g)) return rc;
    if (!pix || !video || (ldp & 3)) return PK_EINVAL;
    const int P = C * pt * ph * pw;
    const long total = (long)B * nt * g.nh * g.nw * (P >> 2);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(unpatchify_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pix, ldp, g, video, total);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

// partials[0 .. PK_SQDIFF_BLOCKS = 1024) <- per-workgroup sums of (a - b)^2 over the frames fmask keeps (NULL: all) of two contiguous
// (B, C, F, H, W) f32 videos, H*W a multiple of 4; the reconstruction-MSE numerator of cvivit.py:585-591.
extern "C" int pk_sqdiff_partials(const float* a, const float* b, const unsigned char* fmask, int B, int C, int F, int H, int W,
                                  double* partials, void* stream) {
    if (!a || !b || !partials || B <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0) return PK_EINVAL;
    if (((long)H * W) & 3) return PK_EALIGN;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return PK_EALIGN;
    const long hw4 = (long)H * W / 4, total4 = (long)B * C * F * hw4;
    hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(1024), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       a, b, fmask, C, F, hw4, total4, partials);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

