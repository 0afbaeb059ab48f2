// Micro-benchmark (round 4): is the L2 -> CU path of a plain global load (buffer_load_dwordx4 -> VGPRs) any wider than that of
// an LDS-DMA (buffer_load_dwordx4 ... lds)?  If the two shared nothing, taking the A operand of the GEMM main loop off the LDS-DMA ring would
// halve the ring's bytes; if they share the CU's one texture-address / L1 path (64 B/clk), the bytes per k-tile are what they are.
//   hipcc --offload-arch=gfx950 -O3 tools/fill_path_bench.hip -o /tmp/fill_path && /tmp/fill_path
// Access pattern = the GEMM operand tiles (8 rows x 128 B per wave-instruction, 1 KiB rows, L2-resident 8 MB matrix).
// MODE 0: every piece by LDS-DMA; 1: every piece to VGPRs (consumed by a v_or chain so the loads cannot be dropped); 2: alternate pieces;
// 3: VGPR loads + ds_write_b128 of the previous piece (the register-staged main loop's traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ void fill_kernel(const char* __restrict__ src, uint32_t bytes, int rows_total, int iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, bytes, 0x00020000);
    char* base = smem + wave * DEPTH * 1024;
    const int groups = rows_total / 8;
    int grp = (blockIdx.x * nw + wave) % groups;
    const uint32_t lane_off = (uint32_t)(lane >> 3) * 1024u + (uint32_t)((lane & 7) ^ (lane >> 3)) * 16u;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = u32x4{0, 0, 0, 0};
    for (int it0 = 0; it0 < iters; it0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int it = it0 + d;
            const int kt = it & 7;
            if (kt == 0 && it) grp = (grp + gridDim.x * nw) % groups;
            const uint32_t voff = (uint32_t)grp * 8192u + lane_off;
            const bool to_lds = MODE == 0 || (MODE == 2 && (d & 1));
            if (to_lds) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(base + d * 1024), 16, voff, kt * 128, 0, 0);
            } else {
                if (MODE == 3) *reinterpret_cast<u32x4*>(base + d * 1024 + lane * 16) = r[d];      // the previous round's piece goes to LDS
                else acc |= r[d];
                r[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, kt * 128, 0);
            }
        }
        // counted wait: the pieces just issued stay in flight (what the GEMM ring does with STAGES - 1 tiles)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc |= r[d];
    if (sink && (acc[0] | acc[1] | acc[2] | acc[3]) == 0x12345678u) sink[0] = acc[0] + reinterpret_cast<uint32_t*>(base)[lane];
}

template <int MODE, int DEPTH>
static double run(const char* src, uint32_t bytes, int rows, int blocks, int threads, int lds_bytes, int iters, uint32_t* sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(blocks), dim3(threads), lds_bytes, 0, src, bytes, rows, iters, sink);
    hipEventRecord(e0, 0);
    for (int rr = 0; rr < 5; ++rr)
        hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(blocks), dim3(threads), lds_bytes, 0, src, bytes, rows, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double total = 5.0 * blocks * (threads / 64) * (double)iters * 1024.0;
    return total / (ms * 1e-3) / 1e12;
}

int main() {
    const int rows = 8192;
    char* small;
    uint32_t* sink;
    hipMalloc(&small, (size_t)rows * 1024);
    hipMalloc(&sink, 4096);
    hipMemset(small, 1, (size_t)rows * 1024);
    const uint32_t bytes = (uint32_t)rows * 1024u;
    const char* names[4] = {"lds-dma", "vgpr", "alternate", "vgpr+ds_write"};
    printf("%-14s %-6s %-6s %-6s %-8s %-12s\n", "mode", "wg/CU", "waves", "depth", "TB/s", "B/clk/CU@2.4G");
    struct Cfg { int wg, waves; };
    std::vector<Cfg> cfgs = {{1, 4}, {2, 4}, {4, 4}, {2, 8}};
    for (auto c : cfgs) {
        const int blocks = 256 * c.wg * 4;
        const int lds = 160 * 1024 / c.wg / 1024 * 1024 - (c.wg > 1 ? 1024 : 0);
        const int iters = 256;
        const int th = c.waves * 64;
        double v[4][2];
        v[0][0] = run<0, 4>(small, bytes, rows, blocks, th, lds, iters, sink); v[0][1] = run<0, 8>(small, bytes, rows, blocks, th, lds, iters, sink);
        v[1][0] = run<1, 4>(small, bytes, rows, blocks, th, lds, iters, sink); v[1][1] = run<1, 8>(small, bytes, rows, blocks, th, lds, iters, sink);
        v[2][0] = run<2, 4>(small, bytes, rows, blocks, th, lds, iters, sink); v[2][1] = run<2, 8>(small, bytes, rows, blocks, th, lds, iters, sink);
        v[3][0] = run<3, 4>(small, bytes, rows, blocks, th, lds, iters, sink); v[3][1] = run<3, 8>(small, bytes, rows, blocks, th, lds, iters, sink);
        for (int m = 0; m < 4; ++m)
            for (int d = 0; d < 2; ++d)
                printf("%-14s %-6d %-6d %-6d %-8.2f %-12.1f\n", names[m], c.wg, c.waves, d ? 8 : 4, v[m][d], v[m][d] * 1e12 / 256 / 2.4e9);
    }
    return 0;
}
