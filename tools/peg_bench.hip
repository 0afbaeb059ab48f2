// PEG (depthwise 3x3x3 + residual) variants timed on the hot-path shapes (standalone, no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/peg_bench.hip -o /tmp/peg_bench && /tmp/peg_bench
// V0 = the product kernel (peg_row_kernel<8>: one thread per (b,t,h,4 channels), 9 conditional (dt,dh) row fetches)
// V1 = same mapping, loads of one dt slab (3 rows x 8 columns) issued together from clamped addresses, taps zeroed by select
// V2 = V1 with ALL 72 loads of the thread's 27-tap window issued before the first FMA (two passes over registers)
// V3 = one thread per (b,t,h,w,4 channels) (8x the threads, 27 loads each, conditional)
#include "../phenaki_pytorch_amd/csrc/elementwise.hip"
#include <cstdio>
#include <vector>

namespace pk {

template <int WW, int GROUP>   // GROUP = number of dt slabs fetched together (1 or 3)
__global__ __launch_bounds__(256) void peg_v1_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     int B, int T, int H, int D, int tfront, long total) {
    const int dv = D >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % dv) * 4;
    long p = idx / dv;
    const int h = (int)(p % H); p /= H;
    const int t = (int)(p % T); const int b = (int)(p / T);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
    const f32x4 zero = f32x4{0, 0, 0, 0};
    f32x4 acc[WW];
#pragma unroll
    for (int w = 0; w < WW; ++w) acc[w] = bv;
#pragma unroll
    for (int d0 = 0; d0 < 3; d0 += GROUP) {
        f32x4 xr[GROUP][3][WW];
        bool ok[GROUP][3];
#pragma unroll
        for (int gi = 0; gi < GROUP; ++gi) {
            const int dt = d0 + gi, ts = t + dt - tfront;
            const int tc = ts < 0 ? 0 : (ts >= T ? T - 1 : ts);
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const int hs = h + dh - 1;
                const int hc = hs < 0 ? 0 : (hs >= H ? H - 1 : hs);
                ok[gi][dh] = ts >= 0 && ts < T && hs >= 0 && hs < H;
                const float* row = x + (((size_t)b * T + tc) * H + hc) * WW * D + c;
#pragma unroll
                for (int w = 0; w < WW; ++w) xr[gi][dh][w] = *reinterpret_cast<const f32x4*>(row + (size_t)w * D);
            }
        }
#pragma unroll
        for (int gi = 0; gi < GROUP; ++gi) {
            const int dt = d0 + gi;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const float* wrow = wt + (size_t)((dt * 3 + dh) * 3) * D + c;
                const f32x4 k0 = ok[gi][dh] ? *reinterpret_cast<const f32x4*>(wrow) : zero;
                const f32x4 k1 = ok[gi][dh] ? *reinterpret_cast<const f32x4*>(wrow + D) : zero;
                const f32x4 k2 = ok[gi][dh] ? *reinterpret_cast<const f32x4*>(wrow + 2 * D) : zero;
#pragma unroll
                for (int w = 0; w < WW; ++w) {
                    if (w > 0) acc[w] += xr[gi][dh][w - 1] * k0;
                    acc[w] += xr[gi][dh][w] * k1;
                    if (w + 1 < WW) acc[w] += xr[gi][dh][w + 1] * k2;
                    if (dt == tfront && dh == 1) acc[w] += xr[gi][dh][w];
                }
            }
        }
    }
    float* orow = out + (((size_t)b * T + t) * H + h) * WW * D + c;
#pragma unroll
    for (int w = 0; w < WW; ++w) *reinterpret_cast<f32x4*>(orow + (size_t)w * D) = acc[w];
}

// V4 = V0's body with an XCD-aware workgroup order: workgroup b runs on XCD b % 8 (private 4 MB L2), so XCD x is given the
// x-th CONTIGUOUS eighth of the (b,t,h) rows -- its 9-row stencil neighbourhoods then hit its own L2 instead of the fabric
template <int WW>
__global__ __launch_bounds__(256) void peg_v4_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     int B, int T, int H, int D, int tfront, long total) {
    const int dv = D >> 2;
    const int per = gridDim.x >> 3;
    const long vb = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const long idx = vb * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % dv) * 4;
    long p = idx / dv;
    const int h = (int)(p % H); p /= H;
    const int t = (int)(p % T); const int b = (int)(p / T);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
    f32x4 acc[WW];
#pragma unroll
    for (int w = 0; w < WW; ++w) acc[w] = bv;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
        const int ts = t + dt - tfront;
        if (ts < 0 || ts >= T) continue;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hs = h + dh - 1;
            if (hs < 0 || hs >= H) continue;
            const float* row = x + (((size_t)b * T + ts) * H + hs) * WW * D + c;
            f32x4 xr[WW];
#pragma unroll
            for (int w = 0; w < WW; ++w) xr[w] = *reinterpret_cast<const f32x4*>(row + (size_t)w * D);
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(wt + (size_t)((dt * 3 + dh) * 3 + 0) * D + c);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(wt + (size_t)((dt * 3 + dh) * 3 + 1) * D + c);
            const f32x4 k2 = *reinterpret_cast<const f32x4*>(wt + (size_t)((dt * 3 + dh) * 3 + 2) * D + c);
#pragma unroll
            for (int w = 0; w < WW; ++w) {
                if (w > 0) acc[w] += xr[w - 1] * k0;
                acc[w] += xr[w] * k1;
                if (w + 1 < WW) acc[w] += xr[w + 1] * k2;
                if (dt == tfront && dh == 1) acc[w] += xr[w];
            }
        }
    }
    float* orow = out + (((size_t)b * T + t) * H + h) * WW * D + c;
#pragma unroll
    for (int w = 0; w < WW; ++w) *reinterpret_cast<f32x4*>(orow + (size_t)w * D) = acc[w];
}

// V5 = V4 with the W-row split in two: one thread per (b,t,h, 4 of the 8 columns, 4 channels) -> twice the threads,
// 6 instead of 8 column loads per row fetch (halo), XCD-contiguous order
template <int WW, int WO>
__global__ __launch_bounds__(256) void peg_v5_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     int B, int T, int H, int D, int tfront, long total) {
    constexpr int NSEG = WW / WO;
    const int dv = D >> 2;
    const long idx = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % dv) * 4;
    long p = idx / dv;
    const int w0 = (int)(p % NSEG) * WO; p /= NSEG;
    const int h = (int)(p % H); p /= H;
    const int t = (int)(p % T); const int b = (int)(p / T);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
    const f32x4 zero = f32x4{0, 0, 0, 0};
    f32x4 acc[WO];
#pragma unroll
    for (int o = 0; o < WO; ++o) acc[o] = bv;
    const bool lv = w0 > 0, rv = w0 + WO < WW;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
        const int ts = t + dt - tfront;
        if (ts < 0 || ts >= T) continue;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hs = h + dh - 1;
            if (hs < 0 || hs >= H) continue;
            const float* row = x + (((size_t)b * T + ts) * H + hs) * WW * D + c;
            f32x4 xr[WO + 2];
            xr[0] = lv ? *reinterpret_cast<const f32x4*>(row + (size_t)(w0 - 1) * D) : zero;
#pragma unroll
            for (int o = 0; o < WO; ++o) xr[o + 1] = *reinterpret_cast<const f32x4*>(row + (size_t)(w0 + o) * D);
            xr[WO + 1] = rv ? *reinterpret_cast<const f32x4*>(row + (size_t)(w0 + WO) * D) : zero;
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(wt + (size_t)((dt * 3 + dh) * 3 + 0) * D + c);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(wt + (size_t)((dt * 3 + dh) * 3 + 1) * D + c);
            const f32x4 k2 = *reinterpret_cast<const f32x4*>(wt + (size_t)((dt * 3 + dh) * 3 + 2) * D + c);
#pragma unroll
            for (int o = 0; o < WO; ++o) {
                acc[o] += xr[o] * k0;
                acc[o] += xr[o + 1] * k1;
                acc[o] += xr[o + 2] * k2;
                if (dt == tfront && dh == 1) acc[o] += xr[o + 1];
            }
        }
    }
    float* orow = out + ((((size_t)b * T + t) * H + h) * WW + w0) * D + c;
#pragma unroll
    for (int o = 0; o < WO; ++o) *reinterpret_cast<f32x4*>(orow + (size_t)o * D) = acc[o];
}

}  // namespace pk

template <typename F>
static float timeit(F launch, int iters = 50) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main() {
    const int D = 512, W = 8;
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int B = cfg == 0 ? 16 : 8, T = 9, H = 8;
        const size_t n = (size_t)B * T * H * W * D;
        float *x, *out, *out2, *wt, *bias;
        hipMalloc(&x, n * 4); hipMalloc(&out, n * 4); hipMalloc(&out2, n * 4); hipMalloc(&wt, 27 * D * 4); hipMalloc(&bias, D * 4);
        std::vector<float> hx(n), hw(27 * D), hb(D);
        for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
        for (int i = 0; i < 27 * D; ++i) hw[i] = (float)((i * 40503u) % 1000) * 1e-3f - 0.5f;
        for (int i = 0; i < D; ++i) hb[i] = 0.01f * i;
        hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(wt, hw.data(), 27 * D * 4, hipMemcpyHostToDevice);
        hipMemcpy(bias, hb.data(), D * 4, hipMemcpyHostToDevice);
        const long rows = (long)B * T * H * (D / 4);
        const dim3 rg((unsigned)((rows + 255) / 256));
        const long tot = (long)B * T * H * W * (D / 4);
        for (int causal = 0; causal < 2; ++causal) {
            const int tf = causal ? 2 : 1;
            const float t0 = timeit([&] { hipLaunchKernelGGL((pk::peg_row_kernel<8>), rg, dim3(256), 0, 0, x, wt, bias, out, B, T, H, D, tf, rows); });
            const float t1 = timeit([&] { hipLaunchKernelGGL((pk::peg_v1_kernel<8, 1>), rg, dim3(256), 0, 0, x, wt, bias, out2, B, T, H, D, tf, rows); });
            std::vector<float> a(n), b2(n);
            hipMemcpy(a.data(), out, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b2.data(), out2, n * 4, hipMemcpyDeviceToHost);
            double e1 = 0; for (size_t i = 0; i < n; ++i) e1 = std::max(e1, (double)fabsf(a[i] - b2[i]));
            const float t2 = timeit([&] { hipLaunchKernelGGL((pk::peg_v1_kernel<8, 3>), rg, dim3(256), 0, 0, x, wt, bias, out2, B, T, H, D, tf, rows); });
            hipMemcpy(b2.data(), out2, n * 4, hipMemcpyDeviceToHost);
            double e2 = 0; for (size_t i = 0; i < n; ++i) e2 = std::max(e2, (double)fabsf(a[i] - b2[i]));
            const float t3 = timeit([&] { hipLaunchKernelGGL(pk::peg_kernel, dim3(16384), dim3(256), 0, 0, x, wt, bias, out2, B, T, H, W, D, tf, tot); });
            const dim3 rg8(8 * ((rg.x + 7) / 8));
            const float t4 = timeit([&] { hipLaunchKernelGGL((pk::peg_v4_kernel<8>), rg8, dim3(256), 0, 0, x, wt, bias, out2, B, T, H, D, tf, rows); });
            hipMemcpy(b2.data(), out2, n * 4, hipMemcpyDeviceToHost);
            double e4 = 0; for (size_t i = 0; i < n; ++i) e4 = std::max(e4, (double)fabsf(a[i] - b2[i]));
            printf("V4 XCD-contiguous order %.1f us (maxdiff %.1e) | ", t4, e4);
            for (int wo : {4, 2}) {
                const long rows5 = rows * (8 / wo);
                const dim3 g5(8 * (((rows5 + 255) / 256 + 7) / 8));
                float t5;
                if (wo == 4) t5 = timeit([&] { hipLaunchKernelGGL((pk::peg_v5_kernel<8, 4>), g5, dim3(256), 0, 0, x, wt, bias, out2, B, T, H, D, tf, rows5); });
                else t5 = timeit([&] { hipLaunchKernelGGL((pk::peg_v5_kernel<8, 2>), g5, dim3(256), 0, 0, x, wt, bias, out2, B, T, H, D, tf, rows5); });
                hipMemcpy(b2.data(), out2, n * 4, hipMemcpyDeviceToHost);
                double e5 = 0; for (size_t i = 0; i < n; ++i) e5 = std::max(e5, (double)fabsf(a[i] - b2[i]));
                printf("V5 split W in %d + XCD order %.1f us (maxdiff %.1e) | ", 8 / wo, t5, e5);
            }
            printf("B=%2d T=%d H=%d W=%d D=%d causal=%d | V0 row kernel %.1f us | V1 slab-of-3 clamped %.1f us (maxdiff %.1e) | V2 all 72 loads first %.1f us (maxdiff %.1e) | V3 thread per token %.1f us | %.1f MB in+out\n",
                   B, T, H, W, D, causal, t0, t1, e1, t2, e2, t3, 2.0 * n * 4 / 1e6);
        }
        hipFree(x); hipFree(out); hipFree(out2); hipFree(wt); hipFree(bias);
    }
    return 0;
}
