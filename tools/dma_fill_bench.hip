// Micro-benchmark: sustained L2/HBM -> LDS fill rate of buffer_load_dwordx4 ... lds on MI355X as a function of
// workgroups per CU, waves per workgroup and DMA pieces in flight per wave.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/dma_fill_bench.hip -o /tmp/dma_fill && /tmp/dma_fill
// Access pattern = the GEMM's operand tiles: each wave-instruction fetches 8 rows x 128 B of a row-major bf16 matrix
// with 1 KiB rows (K = 512), different workgroups walk different row blocks, the k offset advances per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;

template <int DEPTH>   // DMA pieces (1 KiB) each wave keeps in flight
__global__ void fill_kernel(const char* __restrict__ src, uint32_t bytes, int rows_total, int iters, int lds_per_wave, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, bytes, 0x00020000);
    char* base = smem + wave * lds_per_wave;
    // this wave's 8-row group walks through the matrix: row block = (block * nw + wave + t * gridDim.x * nw) % groups
    const int groups = rows_total / 8;
    int grp = (blockIdx.x * nw + wave) % groups;
    const uint32_t lane_off = (uint32_t)(lane >> 3) * 1024u + (uint32_t)((lane & 7) ^ (lane >> 3)) * 16u;
    int issued = 0;
    for (int it = 0; it < iters; ++it) {
        const int kt = it & 7;                                                  // 8 k-tiles of 128 B per 1 KiB row
        if (kt == 0 && it) grp = (grp + gridDim.x * nw) % groups;
        const uint32_t voff = (uint32_t)grp * 8192u + lane_off;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(base + (it % DEPTH) * 1024), 16, voff, kt * 128, 0, 0);
        ++issued;
        if (issued >= DEPTH) {
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && lane == 0 && blockIdx.x == 0xFFFFFF) sink[0] = reinterpret_cast<float*>(base)[0];
}

template <int DEPTH>
static double run(const char* src, uint32_t bytes, int rows, int blocks, int threads, int lds_bytes, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int lds_per_wave = DEPTH * 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fill_kernel<DEPTH>, dim3(blocks), dim3(threads), lds_bytes, 0, src, bytes, rows, iters, lds_per_wave, nullptr);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r)
        hipLaunchKernelGGL(fill_kernel<DEPTH>, dim3(blocks), dim3(threads), lds_bytes, 0, src, bytes, rows, iters, lds_per_wave, nullptr);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = 5.0 * blocks * (threads / 64) * (double)iters * 1024.0;
    return total / (ms * 1e-3) / 1e12;   // TB/s
}

int main() {
    const int rows_small = 8192, rows_big = 512 * 1024;             // 8 MB (L2/MALL resident) and 512 MB (HBM) matrices, 1 KiB rows
    char *small, *big;
    hipMalloc(&small, (size_t)rows_small * 1024);
    hipMalloc(&big, (size_t)rows_big * 1024);
    hipMemset(small, 1, (size_t)rows_small * 1024);
    hipMemset(big, 1, (size_t)rows_big * 1024);
    printf("%-10s %-8s %-6s %-6s %-10s %-10s\n", "footprint", "wg/CU", "waves", "depth", "TB/s", "B/clk/CU@2.4G");
    struct Cfg { int wg_per_cu, waves, depth; };
    std::vector<Cfg> cfgs = {{1, 4, 1}, {1, 4, 4}, {1, 4, 8}, {1, 4, 16}, {2, 4, 4}, {2, 4, 8}, {2, 4, 16}, {4, 4, 2}, {4, 4, 4}, {4, 4, 8},
                             {5, 4, 4}, {8, 4, 1}, {8, 4, 2}, {8, 4, 4}, {1, 8, 8}, {1, 16, 4}, {1, 16, 8}, {2, 8, 8}, {2, 16, 4}};
    for (int foot = 0; foot < 2; ++foot) {
        const char* src = foot ? big : small;
        const int rows = foot ? rows_big : rows_small;
        const uint32_t bytes = foot ? 0x20000000u : (uint32_t)rows_small * 1024u;
        for (auto c : cfgs) {
            const int blocks = 256 * c.wg_per_cu * 4;                            // 4 rounds of resident workgroups
            const int lds = 160 * 1024 / c.wg_per_cu / 1024 * 1024 - (c.wg_per_cu > 1 ? 1024 : 0);   // forces wg_per_cu residency
            const int iters = 256;
            double tbs = 0;
            switch (c.depth) {
                case 1: tbs = run<1>(src, bytes, rows, blocks, c.waves * 64, lds, iters); break;
                case 2: tbs = run<2>(src, bytes, rows, blocks, c.waves * 64, lds, iters); break;
                case 4: tbs = run<4>(src, bytes, rows, blocks, c.waves * 64, lds, iters); break;
                case 8: tbs = run<8>(src, bytes, rows, blocks, c.waves * 64, lds, iters); break;
                default: tbs = run<16>(src, bytes, rows, blocks, c.waves * 64, lds, iters); break;
            }
            printf("%-10s %-8d %-6d %-6d %-10.2f %-10.1f\n", foot ? "512MB" : "8MB", c.wg_per_cu, c.waves, c.depth, tbs, tbs * 1e12 / 256 / 2.4e9);
        }
    }
    return 0;
}
