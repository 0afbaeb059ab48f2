// what does v_dot2c_f32_bf16 compute on gfx950?  hipcc --offload-arch=gfx950 -O3 tools/dot2_check.hip -o /tmp/dot2_check && /tmp/dot2_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* x, float* o) {
    const uint32_t v = x[threadIdx.x];
    const bf2 a = __builtin_bit_cast(bf2, v);
    const bf2 one = {(__bf16)1.0f, (__bf16)1.0f};
    o[threadIdx.x * 3 + 0] = __builtin_amdgcn_fdot2_f32_bf16(a, one, 0.0f, false);
    o[threadIdx.x * 3 + 1] = __builtin_amdgcn_fdot2_f32_bf16(a, a, 0.0f, false);
    o[threadIdx.x * 3 + 2] = __builtin_amdgcn_fdot2_f32_bf16(a, one, 100.0f, false);
}
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
int main() {
    float vals[8][2] = {{1.f, 2.f}, {0.5f, -3.f}, {1.5f, 0.25f}, {-2.f, -4.f}, {100.f, 0.125f}, {0.f, 7.f}, {3.f, 0.f}, {1e-3f, 2e-3f}};
    uint32_t h[8];
    for (int i = 0; i < 8; ++i) h[i] = (uint32_t)f2b(vals[i][0]) | ((uint32_t)f2b(vals[i][1]) << 16);
    uint32_t* dx; float* dout; float out[24];
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, sizeof(out));
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, dx, dout);
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i)
        printf("(%g, %g): dot(a,1)=%g (want %g)  dot(a,a)=%g (want %g)  dot(a,1)+100=%g\n", vals[i][0], vals[i][1], out[i * 3], vals[i][0] + vals[i][1],
               out[i * 3 + 1], vals[i][0] * vals[i][0] + vals[i][1] * vals[i][1], out[i * 3 + 2]);
    return 0;
}
