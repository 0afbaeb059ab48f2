// Micro-benchmark: what the matrix cores of this MI355X SUSTAIN when nothing but MFMAs is issued (register operands, no memory traffic) --
// the practical ceiling behind the 2.5 PFLOP/s (bf16) / 157.3 TFLOP/s (f32) peaks the rooflines are priced against, and the clock it implies.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Per wave: CHAINS independent accumulators, ITER rounds of one MFMA per chain; `waves_per_simd` waves on every SIMD of every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void mfma_kernel(int iters, float* sink) {
    const float seed = (float)(threadIdx.x & 7) * 0.125f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(1.0f - seed); }
    float total = 0.f;
    if (KIND == 0) {                                   // v_mfma_f32_16x16x32_bf16: 16 384 flops
        f32x4 acc[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) total += acc[c][0] + acc[c][3];
    } else if (KIND == 1) {                            // v_mfma_f32_32x32x16_bf16: 32 768 flops
        f32x16 acc[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) total += acc[c][0] + acc[c][15];
    } else {                                           // v_mfma_f32_16x16x4_f32: 2 048 flops
        f32x4 acc[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f - seed, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) total += acc[c][0] + acc[c][3];
    }
    if (total == 12345.678f) sink[0] = total;          // keep the chains alive
}

template <int KIND, int CHAINS>
static void run(const char* name, double flops_per_mfma, int cus, int waves_per_simd, float* sink) {
    const int iters = 20000;
    const int blocks = cus * waves_per_simd;           // 256 threads = 4 waves = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_kernel<KIND, CHAINS>), dim3(blocks), dim3(256), 0, 0, 200, sink);      // warm-up
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_kernel<KIND, CHAINS>), dim3(blocks), dim3(256), 0, 0, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double total = (double)blocks * 4 * iters * CHAINS * flops_per_mfma;
    const double tf = total / (best * 1e-3) / 1e12;
    // cycles per MFMA at full rate: 16x16x32 bf16 = 16 (4 passes x 4), 32x32x16 bf16 = 32, 16x16x4 f32 = 32 -> implied clock
    const double cyc = KIND == 0 ? 16 : 32;
    const double mfma_per_simd_per_s = (double)iters * CHAINS * waves_per_simd / (best * 1e-3);
    printf("%-28s %d wave(s)/SIMD x %d chains: %8.1f TFLOP/s   %.3f ms   implied clock %.2f GHz\n", name, waves_per_simd, CHAINS, tf, best,
           mfma_per_simd_per_s * cyc / 1e9);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clockRate %.2f GHz\n", p.name, p.multiProcessorCount, p.clockRate / 1e6);
    float* sink;
    hipMalloc(&sink, 4);
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0, 8>("bf16 16x16x32", 16384., p.multiProcessorCount, 1, sink); run<1, 4>("bf16 32x32x16", 32768., p.multiProcessorCount, 1, sink); run<2, 8>("f32 16x16x4", 2048., p.multiProcessorCount, 1, sink); }
        if (w == 2) { run<0, 8>("bf16 16x16x32", 16384., p.multiProcessorCount, 2, sink); run<1, 4>("bf16 32x32x16", 32768., p.multiProcessorCount, 2, sink); run<2, 8>("f32 16x16x4", 2048., p.multiProcessorCount, 2, sink); }
        if (w == 4) { run<0, 8>("bf16 16x16x32", 16384., p.multiProcessorCount, 4, sink); run<1, 4>("bf16 32x32x16", 32768., p.multiProcessorCount, 4, sink); run<2, 8>("f32 16x16x4", 2048., p.multiProcessorCount, 4, sink); }
    }
    return 0;
}
