// Where do the microseconds of a short-K pk_gemm launch go?  Standalone timeline probe (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_trace.hip -o /tmp/gemm_trace && /tmp/gemm_trace
// Re-instantiates the product's DMA-ring kernel body (gemm_dma.hpp + the epilogue of gemm.hip) with s_memrealtime stamps
// (100 MHz, chip-global) at workgroup start, after the main loop and after the epilogue; prints per-launch wall time
// (HIP events), the dispatch ramp (when workgroups start), and the per-phase medians, over a sweep of K.
#include "../phenaki_pytorch_amd/csrc/gemm.hip"
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

namespace pk {

template <typename T, int TM, int TN, int WM, int WN, int STAGES, int MODE>
__global__ __launch_bounds__(64 * WM * WN) void traced_kernel(const GemmOperands p, const GemmEpilogue e, int a_nrows, unsigned long long* tr) {
    using Tile = GemmDma<T, TM, TN, WM, WN, STAGES, 128, 0>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long t0 = wall_clock64();
    const int MT = (p.M + Tile::BM - 1) / Tile::BM, cmax = (MT + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int mstart = xcd * MT / 8, mcount = (xcd + 1) * MT / 8 - mstart;
    const int ml = idx % cmax;
    if (ml >= mcount) return;
    const int m0 = (mstart + ml) * Tile::BM, n0 = (idx / cmax) * Tile::BN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    if (MODE != 2) Tile::run(p, a_nrows, m0, n0, smem, acc);
    const unsigned long long t1 = wall_clock64();
    if (MODE != 1) gemm_epilogue<T, TM, TN, WN>(acc, p.M, p.N, e, m0, n0);
    else if (acc[0][0][0] == 12345.678f) gemm_epilogue<T, TM, TN, WN>(acc, p.M, p.N, e, m0, n0);   // keep the loop alive
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tr[blockIdx.x * 4 + 0] = t0; tr[blockIdx.x * 4 + 1] = t1; tr[blockIdx.x * 4 + 2] = t2; tr[blockIdx.x * 4 + 3] = xcc & 15;
    }
}

__global__ void empty_kernel(int) {}

}  // namespace pk

template <int TM, int TN, int WM, int WN, int STAGES, int MODE>
static void run_case(const char* name, int M, int N, int K) {
    using Tile = GemmDma<bf16, TM, TN, WM, WN, STAGES, 128, 0>;
    const int Kp = (K + 63) / 64 * 64;
    void *A, *W, *C; unsigned long long* tr;
    hipMalloc(&A, (size_t)M * Kp * 2); hipMalloc(&W, (size_t)N * Kp * 2); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)M * Kp * 2); hipMemset(W, 0, (size_t)N * Kp * 2);
    const int MT = (M + Tile::BM - 1) / Tile::BM, NT = (N + Tile::BN - 1) / Tile::BN;
    const int grid = 8 * ((MT + 7) / 8) * NT;
    hipMalloc(&tr, (size_t)grid * 32); hipMemset(tr, 0, (size_t)grid * 32);
    GemmOperands p{A, W, nullptr, Kp, Kp, M, N, K, 0, 0};
    GemmEpilogue e{nullptr, nullptr, C, 0, N, 1, ACT_NONE, 1};
    auto kern = traced_kernel<bf16, TM, TN, WM, WN, STAGES, MODE>;
    if (Tile::SMEM > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Tile::SMEM);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(Tile::THREADS), Tile::SMEM, 0, p, e, M, tr);
    hipEventRecord(e0);
    const int IT = 50;
    for (int i = 0; i < IT; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(Tile::THREADS), Tile::SMEM, 0, p, e, M, tr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * 4);
    hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0;
    std::vector<double> start, loop, epi, total;
    int per_xcc[16] = {0};
    for (int b = 0; b < grid; ++b) {
        if (!h[b * 4 + 2]) continue;
        tmin = std::min(tmin, h[b * 4]); tmax = std::max(tmax, h[b * 4 + 2]);
    }
    for (int b = 0; b < grid; ++b) {
        if (!h[b * 4 + 2]) continue;
        start.push_back((h[b * 4] - tmin) * 0.01);
        loop.push_back((h[b * 4 + 1] - h[b * 4]) * 0.01);
        epi.push_back((h[b * 4 + 2] - h[b * 4 + 1]) * 0.01);
        total.push_back((h[b * 4 + 2] - h[b * 4]) * 0.01);
        per_xcc[h[b * 4 + 3] & 15]++;
    }
    auto q = [](std::vector<double>& v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("%-22s M=%d N=%d K=%4d wgs=%4zu  launch %.2f us | span %.2f us | start p50 %.2f p90 %.2f max %.2f | loop p50 %.2f p90 %.2f | "
           "epi p50 %.2f p90 %.2f | wg total p50 %.2f max %.2f | xcc", name, M, N, K, start.size(), ms * 1e3 / IT, (tmax - tmin) * 0.01,
           q(start, .5), q(start, .9), q(start, 1.0), q(loop, .5), q(loop, .9), q(epi, .5), q(epi, .9), q(total, .5), q(total, 1.0));
    for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
    printf("\n");
    hipFree(A); hipFree(W); hipFree(C); hipFree(tr);
}

int main() {
    {   // launch floor: back-to-back empty kernels of the same grid
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int g : {1, 576, 2304}) {
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(pk::empty_kernel, dim3(g), dim3(256), 0, 0, 0);
            hipEventRecord(e0);
            for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(pk::empty_kernel, dim3(g), dim3(256), 0, 0, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("empty kernel grid %4d x 256: %.2f us per launch\n", g, ms * 10);
        }
    }
    for (int K : {64, 128, 256, 512, 1024, 2048}) run_case<2, 2, 2, 2, 2, 0>("64x64 s2 full", 4608, 512, K);
    for (int K : {64, 512, 2048}) run_case<2, 2, 2, 2, 2, 1>("64x64 s2 loop-only", 4608, 512, K);
    run_case<2, 2, 2, 2, 2, 2>("64x64 s2 epi-only", 4608, 512, 512);
    for (int K : {512, 2048}) run_case<2, 2, 2, 2, 4, 0>("64x64 s4 full", 4608, 512, K);
    for (int K : {512, 2048}) run_case<4, 2, 2, 4, 2, 0>("128x128 w8 s2 full", 4608, 512, K);
    for (int K : {512}) run_case<2, 2, 2, 2, 2, 0>("64x64 s2 full", 9216, 2736, K);
    for (int K : {512}) run_case<4, 2, 2, 4, 2, 0>("128x128 w8 s2 full", 9216, 2736, K);
    for (int K : {512}) run_case<4, 2, 2, 4, 2, 1>("128x128 w8 s2 loop", 9216, 2736, K);
    return 0;
}
