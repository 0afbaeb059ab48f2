// Training-mode auxiliary loss of the lookup-free quantizer (reference cvivit.py:570 third return of `self.vq`, added to the
// GAN generator objective at cvivit.py:666; the quantizer itself is the un-vendored vector-quantize-pytorch LFQ, restated in oracle/lfq.py):
//
//     prob_i  = softmax_c( 2 T <z_i, code_c> )                over the 2^cd sign codes,  z_i = project_in(x_i)  (cd values per token)
//     aux     = w_e * ( mean_i H(prob_i)  -  gamma * H(mean_i prob_i) )  +  w_c * mean (z - sign(z) s)^2,     H(p) = sum -p log(max(p, 1e-5))
//
// The published module materialises prob as an (M, 2^cd) matrix (1.2 GB at M = 4 608, cd = 16).  It never has to exist: the codes are all sign
// combinations, so the softmax FACTORISES over the bits -- prob_i(c) = prod_k sigma(+-alpha z_ik), alpha = 4 T s -- and with the bits split into a
// high and a low half, prob_i(u, v) = a_i[u] * b_i[v] with 2^(cd/2)-entry vectors a_i, b_i:
//   * per-token entropy: a workgroup owns one token and walks the 2^cd products a[u] b[v] in registers / LDS (pk_lfq_aux_prep);
//   * batch distribution: mean_i prob_i = A^T B / M, a (2^hi x 2^lo x M) product of the (M, 2^hi) and (M, 2^lo) factor matrices, which DO fit
//     (4.7 MB each) -- run by the exact-f32 pk_bmm; its entropy and the derivative wrt every entry in pk_lfq_aux_codebook;
//   * gradient wrt z: the chain back through A^T B is two more small products (B G^T, A G), the per-token part is walked again, and both fold
//     onto the cd logits through d a[u] / d z_k = alpha a[u] (bit_k(u) - sigma_k) (pk_lfq_aux_grad); the clamp in H keeps torch's derivative
//     (0 through the clamped branch: d/dp [-p log max(p, eps)] = -log eps for p < eps, -log p - 1 otherwise).
// Everything is f32 with libm logf / expf: the loss is a scalar summed over 3e8 terms, not a throughput kernel (~60 us at BASELINE size).
#include "common.hpp"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace {

constexpr float LOG_EPS_P = 1e-5f;                    // the published `log(t, eps = 1e-5)`: t.clamp(min = eps).log()
constexpr float NEG_LOG_EPS = 11.512925464970229f;    // -log(1e-5)

__device__ __forceinline__ float ent_term(float p) { return p >= LOG_EPS_P ? -p * logf(p) : p * NEG_LOG_EPS; }
__device__ __forceinline__ float ent_deriv(float p) { return p >= LOG_EPS_P ? -(logf(p) + 1.0f) : NEG_LOG_EPS; }

// sum over the 256 threads of a workgroup (all threads get the result); `red` = 4 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// the token's bit probabilities p_k = sigma(alpha z_k) (bit k set) and q_k = sigma(-alpha z_k), then the factor vectors a (high bits) / b (low bits).
// Code index c = sum_k bit_k 2^(cd-1-k) (MSB first, the `mask` buffer of the module): u = c >> lo holds dims 0..hi-1, v = c & (2^lo - 1) the rest.
__device__ __forceinline__ void token_factors(const float* __restrict__ z, int cd, int hi, int lo, float alpha, float* sp, float* sq, float* sa, float* sb) {
    const int t = threadIdx.x;
    if (t < cd) {
        const float x = alpha * z[t];
        const float e = expf(-fabsf(x));
        const float big = 1.0f / (1.0f + e), small = e / (1.0f + e);
        sp[t] = x >= 0.f ? big : small;
        sq[t] = x >= 0.f ? small : big;
    }
    __syncthreads();
    if (t < (1 << hi)) {
        float a = 1.0f;
        for (int k = 0; k < hi; ++k) a *= ((t >> (hi - 1 - k)) & 1) ? sp[k] : sq[k];
        sa[t] = a;
    }
    if (t < (1 << lo)) {
        float b = 1.0f;
        for (int k = hi; k < cd; ++k) b *= ((t >> (cd - 1 - k)) & 1) ? sp[k] : sq[k];
        sb[t] = b;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void lfq_aux_prep_kernel(const float* __restrict__ proj, int cd, float alpha, float scale, float* __restrict__ A,
                                                           float* __restrict__ B, float* __restrict__ ent, float* __restrict__ commit) {
    __shared__ float sp[16], sq[16], sa[256], sb[256], red[4];
    const int i = blockIdx.x, t = threadIdx.x;
    const int hi = (cd + 1) >> 1, lo = cd >> 1, NA = 1 << hi, NB = 1 << lo;
    const float* z = proj + (size_t)i * cd;
    token_factors(z, cd, hi, lo, alpha, sp, sq, sa, sb);
    if (t < NA) A[(size_t)i * NA + t] = sa[t];
    if (t < NB) B[(size_t)i * NB + t] = sb[t];
    float h = 0.f;
    if (t < NA) {
        const float a = sa[t];
        for (int v = 0; v < NB; ++v) h += ent_term(a * sb[v]);
    }
    h = block_sum(h, red);
    float c = 0.f;
    if (t < cd) {
        const float d = z[t] - (z[t] > 0.f ? scale : -scale);
        c = d * d;
    }
    c = block_sum(c, red);
    if (t == 0) { ent[i] = h; commit[i] = c; }
}

// Q = A^T B (sums over the tokens): q = Q * inv_n is the batch distribution.  hc_part[block] = partial sum of ent_term(q); G = coef * ent_deriv(q)
__global__ __launch_bounds__(256) void lfq_aux_codebook_kernel(const float* __restrict__ Q, int total, float inv_n, float coef, float* __restrict__ G,
                                                               float* __restrict__ hc_part) {
    __shared__ float red[4];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float h = 0.f;
    if (idx < total) {
        const float q = Q[idx] * inv_n;
        h = ent_term(q);
        G[idx] = coef * ent_deriv(q);
    }
    h = block_sum(h, red);
    if (threadIdx.x == 0) hc_part[blockIdx.x] = h;
}

// d aux / d z for one token: GA (M, NA) / GB (M, NB) = the batch-entropy term's gradient wrt a / b (already scaled), wen = w_e / M the weight of this
// token's own entropy, wc2 = 2 w_c / (M cd) the commitment term's
__global__ __launch_bounds__(256) void lfq_aux_grad_kernel(const float* __restrict__ proj, const float* __restrict__ GA, const float* __restrict__ GB, int cd,
                                                           float alpha, float scale, float wen, float wc2, float* __restrict__ dproj) {
    __shared__ float sp[16], sq[16], sa[256], sb[256], ta[256], tb[256];
    const int i = blockIdx.x, t = threadIdx.x;
    const int hi = (cd + 1) >> 1, lo = cd >> 1, NA = 1 << hi, NB = 1 << lo;
    const float* z = proj + (size_t)i * cd;
    token_factors(z, cd, hi, lo, alpha, sp, sq, sa, sb);
    if (t < NA) {
        const float a = sa[t];
        float acc = 0.f;
        for (int v = 0; v < NB; ++v) acc = fmaf(ent_deriv(a * sb[v]), sb[v], acc);
        ta[t] = (GA[(size_t)i * NA + t] + wen * acc) * a;                    // (d aux / d a_u) * a_u
    }
    if (t < NB) {
        const float b = sb[t];
        float acc = 0.f;
        for (int u = 0; u < NA; ++u) acc = fmaf(ent_deriv(sa[u] * b), sa[u], acc);
        tb[t] = (GB[(size_t)i * NB + t] + wen * acc) * b;
    }
    __syncthreads();
    if (t < cd) {
        // d a_u / d z_k = alpha a_u (bit_k(u) - p_k):  bit set -> alpha a_u q_k,  clear -> -alpha a_u p_k
        float s1 = 0.f, s0 = 0.f;
        if (t < hi) {
            const int sh = hi - 1 - t;
            for (int u = 0; u < NA; ++u) { if ((u >> sh) & 1) s1 += ta[u]; else s0 += ta[u]; }
        } else {
            const int sh = cd - 1 - t;
            for (int v = 0; v < NB; ++v) { if ((v >> sh) & 1) s1 += tb[v]; else s0 += tb[v]; }
        }
        const float zz = z[t];
        dproj[(size_t)i * cd + t] = alpha * (sq[t] * s1 - sp[t] * s0) + wc2 * (zz - (zz > 0.f ? scale : -scale));
    }
}

// out[0] = aux, out[1] = per-sample entropy, out[2] = codebook entropy, out[3] = commitment; one workgroup, fixed summation order
__global__ __launch_bounds__(256) void lfq_aux_finish_kernel(const float* __restrict__ ent, const float* __restrict__ commit, int M, const float* __restrict__ hc_part,
                                                             int nparts, float w_e, float gamma, float w_c, int cd, float* __restrict__ out) {
    __shared__ float red[4];
    float e = 0.f, c = 0.f, h = 0.f;
    for (int i = threadIdx.x; i < M; i += 256) { e += ent[i]; c += commit[i]; }
    for (int i = threadIdx.x; i < nparts; i += 256) h += hc_part[i];
    e = block_sum(e, red);
    c = block_sum(c, red);
    h = block_sum(h, red);
    if (threadIdx.x == 0) {
        const float pe = e / (float)M, cm = c / ((float)M * (float)cd);
        out[0] = w_e * (pe - gamma * h) + w_c * cm;
        out[1] = pe; out[2] = h; out[3] = cm;
    }
}

inline bool bad_cd(int cd) { return cd < 2 || cd > 16; }

}  // namespace

extern "C" int pk_lfq_aux_parts(int cd) { return bad_cd(cd) ? 0 : ((1 << cd) + 255) / 256; }

extern "C" int pk_lfq_aux_prep(const float* proj, int M, int cd, float alpha, float scale, float* A, float* B, float* ent, float* commit, void* stream) {
    if (!proj || !A || !B || !ent || !commit || M <= 0 || bad_cd(cd)) return PK_EINVAL;
    hipLaunchKernelGGL(lfq_aux_prep_kernel, dim3(M), dim3(256), 0, STREAM(stream), proj, cd, alpha, scale, A, B, ent, commit);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_lfq_aux_codebook(const float* Q, int cd, float inv_n, float coef, float* G, float* hc_part, void* stream) {
    if (!Q || !G || !hc_part || bad_cd(cd)) return PK_EINVAL;
    const int total = 1 << cd;
    hipLaunchKernelGGL(lfq_aux_codebook_kernel, dim3((total + 255) / 256), dim3(256), 0, STREAM(stream), Q, total, inv_n, coef, G, hc_part);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_lfq_aux_grad(const float* proj, const float* GA, const float* GB, int M, int cd, float alpha, float scale, float wen, float wc2,
                               float* dproj, void* stream) {
    if (!proj || !GA || !GB || !dproj || M <= 0 || bad_cd(cd)) return PK_EINVAL;
    hipLaunchKernelGGL(lfq_aux_grad_kernel, dim3(M), dim3(256), 0, STREAM(stream), proj, GA, GB, cd, alpha, scale, wen, wc2, dproj);
    PK_CHECK_LAUNCH();
    return PK_OK;
}

extern "C" int pk_lfq_aux_finish(const float* ent, const float* commit, int M, const float* hc_part, int cd, float w_e, float gamma, float w_c, float* out,
                                 void* stream) {
    if (!ent || !commit || !hc_part || !out || M <= 0 || bad_cd(cd)) return PK_EINVAL;
    hipLaunchKernelGGL(lfq_aux_finish_kernel, dim3(1), dim3(256), 0, STREAM(stream), ent, commit, M, hc_part, ((1 << cd) + 255) / 256, w_e, gamma, w_c, cd, out);
    PK_CHECK_LAUNCH();
    return PK_OK;
}
