// Shared device helpers for the gfx950 (MI355X / CDNA4) Phenaki hot-path kernels.
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, written as the literal 64 everywhere;
//   * "fragment chunk" = a 16 x 32 (row x k) slice of an MFMA operand held by one wave:
//       lane l holds row (l & 15), logical k = (l >> 4) * 8 + j, j = 0..7  (8 elements per lane).
//     For bf16 that is exactly one v_mfma_f32_16x16x32_bf16 operand; for exact-f32 mode the same
//     8 elements feed eight v_mfma_f32_16x16x4_f32 (step j uses k-set {g*8+j : g = 0..3}).  Both
//     operands of a product always use the SAME lane->k map, so any such permutation is exact.
//   * accumulators follow the gfx950 C/D map: reg r of lane l = D[row = (l >> 4) * 4 + r][col = l & 15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct bf16 { u16 v; };   // storage-only bf16 (round-to-nearest-even from f32, like torch)

// ---- split-bf16 ("bf16x3") operand types: f32-grade products on the bf16 matrix cores -----------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest-even): |x - (hi + lo)| <= 2^-18 |x|.  A product of two such
// operands is accumulated as hi.hi + hi.lo + lo.hi (three v_mfma_f32_16x16x32_bf16 into the same f32 accumulator; the dropped lo.lo
// term is <= 2^-18 of the product), i.e. ~1e-5 per-product relative error against 2^-9 for plain bf16 operands -- enough to hold the
// reference's f32 results to the north-star tolerances (1e-3, bit-exact ids) at 3/16 of the f32-MFMA cost.
//   bf16x3  : an f32 value in memory (every load / store / epilogue treats it as float).  As a GEMM operand type it means: A rows are
//             f32 in HBM and LDS and are split in registers after the ds_read; W is PRE-SPLIT by the host packer into 128-byte blocks
//             of 32 k-elements, [hi x 32 | lo x 32] bf16 -- the same bytes per row as f32, so the LDS-DMA ring is unchanged.
//   bf16x3p : an element of such a pre-split image IN MEMORY (the attention operand images Q^ / K^ / V^T written by pk_attn_prep):
//             element e of a row lives in block e / 32 as hi at byte (e % 32) * 2 and lo at byte 64 + (e % 32) * 2.  sizeof = 4 and
//             rows are multiples of 32 elements from a 128-byte aligned base, so ordinary `T* + element index` pointer arithmetic
//             still lands in the right block: the accessors below recover (block, position) from the address itself.
struct bf16x3 { float v; };
struct bf16x3p { uint32_t v; };

// f32 -> bf16 through the native __bf16 type: hipcc lowers it to v_cvt_pk_bf16_f32 on gfx950 (hardware
// round-to-nearest-even, one instruction per PAIR instead of ~5 VALU per element for a manual RNE)
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16 f2bf(float f) {
    return __builtin_bit_cast(u16, (__bf16)f);
}
__device__ __forceinline__ float bf2f(u16 h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const bf16x2_native v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// the in-memory element type of the attention operand images (Q^ / K^ / V^T) a GEMM operand type writes: bf16 -> bf16, bf16x3 -> bf16x3p
template <typename T> struct ImageOf { typedef T type; };
template <> struct ImageOf<bf16x3> { typedef bf16x3p type; };

// ---- per-type fragment chunk -------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16> { u32x4 v; };                 // 8 bf16
template <> struct Frag<float> { f32x4 lo, hi; };           // 8 f32 (k = g*8 + 0..3 | 4..7)
template <> struct Frag<bf16x3> { u32x4 hi, lo; };          // 8 values as two bf16 planes: value = hi + lo
template <> struct Frag<bf16x3p> { u32x4 hi, lo; };

// acc += A_chunk (rows i) x B_chunk (cols j) over the chunk's 32 logical k
__device__ __forceinline__ f32x4 mma(const Frag<bf16>& a, const Frag<bf16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v),
                                                    __builtin_bit_cast(bf16x8_t, b.v), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Frag<float>& a, const Frag<float>& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], c, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], c, 0, 0, 0);
    return c;
}

// split-bf16: 3 MFMAs per fragment pair, small cross terms first
template <typename TS>
__device__ __forceinline__ f32x4 mma_split(const Frag<TS>& a, const Frag<TS>& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.hi), __builtin_bit_cast(bf16x8_t, b.lo), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.lo), __builtin_bit_cast(bf16x8_t, b.hi), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.hi), __builtin_bit_cast(bf16x8_t, b.hi), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Frag<bf16x3>& a, const Frag<bf16x3>& b, f32x4 c) { return mma_split(a, b, c); }
__device__ __forceinline__ f32x4 mma(const Frag<bf16x3p>& a, const Frag<bf16x3p>& b, f32x4 c) { return mma_split(a, b, c); }

// 8 f32 values -> (hi, lo) bf16 planes: 4 v_cvt_pk (hi, RNE) + 8 shift/and (hi back to f32) + 8 v_sub (exact: Sterbenz-like, the
// difference of an f32 and its own 8-bit-mantissa rounding fits in 16 bits) + 4 v_cvt_pk (lo) = 24 VALU per fragment
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t h = pack_bf2(x[2 * i], x[2 * i + 1]);
        hi[i] = h;
        const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xFFFF0000u);
        lo[i] = pack_bf2(x[2 * i] - h0, x[2 * i + 1] - h1);
    }
}

// 8 contiguous elements at p (16-byte aligned for bf16, 32-byte span for f32) -> fragment chunk
__device__ __forceinline__ void frag_load(Frag<bf16>& f, const bf16* p) {
    f.v = *reinterpret_cast<const u32x4*>(p);
}
__device__ __forceinline__ void frag_load(Frag<float>& f, const float* p) {
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 4);
}
// pre-split image: p addresses element e of a row (e % 8 == 0); the 8 values are hi at block + (e % 32) * 2, lo 64 bytes further
__device__ __forceinline__ const char* split_block(const void* p, int& pos2) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    pos2 = (int)(a & 127) >> 1;                                  // (e % 32) * 2
    return reinterpret_cast<const char*>(a & ~(uintptr_t)127);
}
__device__ __forceinline__ void frag_load(Frag<bf16x3p>& f, const bf16x3p* p) {
    int o;
    const char* blk = split_block(p, o);
    f.hi = *reinterpret_cast<const u32x4*>(blk + o);
    f.lo = *reinterpret_cast<const u32x4*>(blk + 64 + o);
}
__device__ __forceinline__ void frag_zero(Frag<bf16x3p>& f) { f.hi = u32x4{0, 0, 0, 0}; f.lo = f.hi; }
__device__ __forceinline__ void frag_zero(Frag<bf16x3>& f) { f.hi = u32x4{0, 0, 0, 0}; f.lo = f.hi; }
__device__ __forceinline__ void frag_zero(Frag<bf16>& f) { f.v = u32x4{0, 0, 0, 0}; }
__device__ __forceinline__ void frag_zero(Frag<float>& f) { f.lo = f32x4{0, 0, 0, 0}; f.hi = f.lo; }

// build a fragment chunk from 8 f32 values (register-resident P of the attention kernel)
__device__ __forceinline__ void frag_from_f32(Frag<bf16>& f, const float (&x)[8]) {
    f.v = u32x4{pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7])};
}
__device__ __forceinline__ void frag_from_f32(Frag<float>& f, const float (&x)[8]) {
    f.lo = f32x4{x[0], x[1], x[2], x[3]};
    f.hi = f32x4{x[4], x[5], x[6], x[7]};
}

__device__ __forceinline__ void frag_from_f32(Frag<bf16x3p>& f, const float (&x)[8]) {
    split8(f32x4{x[0], x[1], x[2], x[3]}, f32x4{x[4], x[5], x[6], x[7]}, f.hi, f.lo);
}
__device__ __forceinline__ void frag_from_f32(Frag<bf16x3>& f, const float (&x)[8]) {
    split8(f32x4{x[0], x[1], x[2], x[3]}, f32x4{x[4], x[5], x[6], x[7]}, f.hi, f.lo);
}

// scalar store/convert helpers
__device__ __forceinline__ void store_elem(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_elem(bf16* p, float v) { p->v = f2bf(v); }
__device__ __forceinline__ float load_elem(const float* p) { return *p; }
__device__ __forceinline__ float load_elem(const bf16* p) { return bf2f(p->v); }

__device__ __forceinline__ void store_elem(bf16x3* p, float v) { p->v = v; }
__device__ __forceinline__ float load_elem(const bf16x3* p) { return p->v; }
__device__ __forceinline__ void store4(bf16x3* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store2(bf16x3* p, float a, float b) { *reinterpret_cast<f32x2*>(p) = f32x2{a, b}; }
// 4 consecutive elements (e % 4 == 0) of a pre-split image: 8 bytes into each plane of the block the address falls into
__device__ __forceinline__ void store4(bf16x3p* p, f32x4 v) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    char* blk = reinterpret_cast<char*>(a & ~(uintptr_t)127);
    const int o = (int)(a & 127) >> 1;
    const uint32_t h0 = pack_bf2(v[0], v[1]), h1 = pack_bf2(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h0 << 16), r1 = v[1] - __builtin_bit_cast(float, h0 & 0xFFFF0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h1 << 16), r3 = v[3] - __builtin_bit_cast(float, h1 & 0xFFFF0000u);
    *reinterpret_cast<u32x2*>(blk + o) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(blk + 64 + o) = u32x2{pack_bf2(r0, r1), pack_bf2(r2, r3)};
}
// one element of a pre-split image (the V^T scatter of the projection kernels): hi and lo as two 2-byte stores into the block's planes
__device__ __forceinline__ void store_elem(bf16x3p* p, float v) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    char* blk = reinterpret_cast<char*>(a & ~(uintptr_t)127);
    const int o = (int)(a & 127) >> 1;
    const u16 h = f2bf(v);
    *reinterpret_cast<u16*>(blk + o) = h;
    *reinterpret_cast<u16*>(blk + 64 + o) = f2bf(v - bf2f(h));
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, f32x4 v) {
    *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}
__device__ __forceinline__ void store2(float* p, float a, float b) { *reinterpret_cast<f32x2*>(p) = f32x2{a, b}; }
__device__ __forceinline__ void store2(bf16* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_bf2(a, b); }

// ---- XCD-contiguous workgroup order -------------------------------------------------------
// Workgroup b of a launch is observed to run on XCD b % 8, each XCD with a private 4 MB L2 (speed only, never
// correctness).  A kernel whose NEIGHBOURING workgroups share data (the PEG stencil rows) launches 8 * per workgroups and
// walks them in this order, so XCD x owns the x-th contiguous eighth of the work and the shared rows are fetched over the
// fabric once instead of (up to) eight times.  vb >= the real count: exit.  (Measured neutral, and not used, for the
// q-blocks of one attention head and for the vocab-head tiles: their shared operands are served by the Infinity Cache.)
__device__ __forceinline__ long xcd_contiguous_block(unsigned b, unsigned nblocks_padded) {
    return (long)(b & 7u) * (nblocks_padded >> 3) + (b >> 3);
}
__host__ __device__ inline unsigned xcd_padded_grid(long nblocks) { return (unsigned)(8 * ((nblocks + 7) / 8)); }

// ---- wave reductions (64 lanes) ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exact erf GELU (torch F.gelu default)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// the same GELU for results that are rounded to bf16 right away (the GEGLU epilogue in bf16 mode): erf by Abramowitz-Stegun
// 7.1.26 -- |error| <= 1.5e-7 absolute, 3 to 4 decimal digits below a bf16 ulp of the result -- with one v_rcp and one v_exp
// instead of libm's branchy erff (the epilogue of the FF1 GEMM was 2.1 of 9 us per workgroup, profiles/gemm_timeline_r01.txt)
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfz = fmaf(-p * t, __expf(-z * z), 1.0f);          // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(erfz, x));
}
template <typename T> __device__ __forceinline__ float gelu_for(float x);
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_for<bf16>(float x) { return gelu_erf_fast(x); }
// split-bf16: the products in front of this GELU carry ~1e-5 relative error; the 1.5e-7 absolute error of the rational erf is two orders below that, and libm's branchy
// erff was 20 % of the FF1 launch in that mode (round 6 A/B, profiles/parity_gelu_r06.txt).  PK_X3_EXACT_GELU: the round-5 behaviour, for that A/B.
#ifdef PK_X3_EXACT_GELU
template <> __device__ __forceinline__ float gelu_for<bf16x3>(float x) { return gelu_erf(x); }
#else
template <> __device__ __forceinline__ float gelu_for<bf16x3>(float x) { return gelu_erf_fast(x); }
#endif

// ---- counter-based uniform noise shared by the sampler kernels and their tests -------------
// u = hash(seed, stream, index) mapped to [0, 1) with 24 bits, the same granularity torch's
// float uniform_ has.  tests/ re-implement this in numpy to feed the identical noise to the oracle.
__device__ __host__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __host__ __forceinline__ float uniform24(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx_lo, uint32_t idx_hi) {
    uint32_t h = mix32(idx_lo ^ seed_lo);
    h = mix32(h + 0x9e3779b9u * (idx_hi + 1u) + seed_hi);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// four draws for the 4 consecutive elements of group q = flat_index >> 2 (the vocab head's epilogue owns 4 consecutive
// columns per lane): ONE hash chain per group, three finalisers = 96 bits = 4 x 24 -- half the integer multiplies of
// uniform24 per element (v_mul_lo_u32 is a quarter-rate instruction and the hash was the largest part of the epilogue)
__device__ __host__ __forceinline__ void uniform24x4(uint32_t seed_lo, uint32_t seed_hi, uint32_t q_lo, uint32_t q_hi, float (&u)[4]) {
    uint32_t h = mix32(q_lo ^ seed_lo);
    h = h + 0x9e3779b9u * (q_hi + 1u) + seed_hi;
    const uint32_t w0 = mix32(h), w1 = mix32(h + 0x85EBCA77u), w2 = mix32(h + 0x0BD794EEu);
    const float k = 1.0f / 16777216.0f;
    u[0] = (float)(w0 >> 8) * k;
    u[1] = (float)(w1 >> 8) * k;
    u[2] = (float)(w2 >> 8) * k;
    u[3] = (float)(((w0 & 0xFFu) << 16) | ((w1 & 0xFFu) << 8) | (w2 & 0xFFu)) * k;
}

// ---- torch's device RNG, reproduced (SURVEY.md 7 "phase 2"; reference phenaki_pytorch.py:69-70, :88-93: noise = zeros_like(t).uniform_(0, 1)) ----
// torch.Tensor.uniform_ on a HIP device fills element li of a tensor of `numel` elements from Philox4x32-10 (rocRAND) as follows
// (ATen/native/cuda/DistributionTemplates.h: distribution_elementwise_grid_stride_kernel, block 256, unroll 4):
//   grid   = min(#CUs * (max threads per CU / 256), ceil(numel / 256)),   stride = 256 * grid
//   thread = li % stride (the Philox SUBSEQUENCE),  call k = li / (4 stride) (the thread's k-th 4-word draw),  word = (li / stride) % 4
//   x      = philox4x32_10(counter = {offset / 4 + k (64 bit), thread (64 bit)}, key = seed)[word]
//   u      = 2^-32 + float(x) * 2^-32   (rocrand_uniform),   u == 1 -> 0   (uniform_ maps (0, 1] to [0, 1))
// and the generator's offset then advances by 4 * ceil(numel / (4 stride)).  Bit-exact against torch.zeros(n, device='cuda').uniform_()
// on MI355X (tests/test_kernels_gpu.py).  Used by the PARITY sampler so that a seeded reference run on the same GPU is reproducible.
__device__ __forceinline__ uint32_t philox4x32_10_word(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int word) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return word == 0 ? c0 : word == 1 ? c1 : word == 2 ? c2 : c3;
}
__device__ __forceinline__ float torch_uniform(uint32_t seed_lo, uint32_t seed_hi, uint64_t offset, uint64_t li, uint32_t stride) {
    const uint64_t q = li / stride;
    const uint32_t thread = (uint32_t)(li - q * stride);
    const uint64_t n = (offset >> 2) + (q >> 2);
    const uint32_t x = philox4x32_10_word((uint32_t)n, (uint32_t)(n >> 32), thread, 0u, seed_lo, seed_hi, (int)(q & 3));
    const float u = 2.3283064365386963e-10f + (float)x * 2.3283064365386963e-10f;
    return u == 1.0f ? 0.0f : u;
}

constexpr float NEG_MAX = -3.402823466e+38f;   // -finfo(float32).max, the reference's mask fill value

}  // namespace pk

#define PK_OK 0
#define PK_EINVAL (-1)       // bad shape / size argument
#define PK_EALIGN (-2)       // pointer or stride alignment violated
#define PK_ELAUNCH (-3)      // hipGetLastError() after the launch was not hipSuccess

#define PK_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return PK_ELAUNCH; } while (0)
