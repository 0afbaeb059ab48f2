/* phenaki_hip.h -- C ABI of libphenaki_hip.so, the MI355X (gfx950) kernels behind the Phenaki hot path.
 *
 * The reference (lucidrains/phenaki-pytorch) has no FFI: its boundary is the Python nn.Module surface
 * (CViViT / MaskGit / TokenCritic / Phenaki).  This library is what those modules' forward passes bind
 * in the MI355X build (phenaki_pytorch_amd/_lib.py, ctypes); every entry point names the reference
 * expression it replaces (paths relative to /root/reference/phenaki_pytorch/).
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller; kernels never allocate or free;
 *   - `stream` is a hipStream_t (0 = default stream); every call is asynchronous on it, no hidden syncs;
 *   - return value: 0 on success, negative on error (no exceptions cross this boundary):
 *       PK_EINVAL (-1) bad shape/size/flag, PK_EALIGN (-2) pointer/stride alignment, PK_ELAUNCH (-3) HIP launch error;
 *   - `dtype`: 0 = exact f32 (f32 MFMA, bit-for-bit an fmaf chain), 1 = bf16 MFMA inputs with f32 accumulation.
 *     "T" below means float for dtype 0 and bf16 (uint16 storage, round-to-nearest-even) for dtype 1;
 *     2 = split-bf16 ("bf16x3", pk_gemm[_ex] / pk_attn_prep / pk_attn_fwd / pk_vocab_sample / pk_vocab_ce): every product is
 *     accumulated as hi.hi + hi.lo + lo.hi on the bf16 matrix cores with x = hi + lo, hi = bf16(x), lo = bf16(x - hi)
 *     (~1e-5 per product; held to the f32 tolerances).  Activations stay f32 (T = float, a_is_f32 / out_is_f32 = 1);
 *     W [N][ldw] is the PRE-SPLIT image of the zero-padded f32 weight: per row and per block of 32 k-elements 128 bytes =
 *     [hi x 32 | lo x 32] bf16 (ldw counted in 4-byte units, a multiple of 32; _lib.split_planes); the attention operand
 *     images Qp / Kp / Vt use the same blocked (hi | lo) format per row (same sizes as dtype 0, 128-byte aligned bases).
 *     LDS-DMA main loops only; LayerNorm fold (ln_s / ln_t), row statistics (stats_out / ln_stats) and dup_rows work as for dtype 1
 *     (round 4; the rows ARE f32, so no C2 copy exists or is needed);
 *   - activations in HBM are f32 unless a parameter says T; rows are row-major with an explicit leading
 *     dimension (ld*, in elements);
 *   - dim_head is fixed at 64 in the attention kernels.
 */
#ifndef PHENAKI_HIP_H
#define PHENAKI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PK_OK 0
#define PK_EINVAL (-1)
#define PK_EALIGN (-2)
#define PK_ELAUNCH (-3)

/* C = act(A @ W^T + bias) (+ res).  Replaces every nn.Linear on the path: attention.py:50-52 (FeedForward,
 * act 1 = GEGLU on interleaved (value, gate) column pairs -> C is [M][N/2]), attention.py:117-119 (to_q/to_kv/to_out),
 * attention.py:243-252 (CPB MLP, act 2 = LeakyReLU(0.1)), cvivit.py:276,283 (patch embed), cvivit.py:328,333
 * (to_pixels), phenaki_pytorch.py:147 (to_logits).
 * A [M][lda] is f32 (a_is_f32) or T; W [N][ldw] is T; bias [N] f32 or NULL; res [M][ldr] f32 or NULL;
 * C is f32 (out_is_f32) or T; a_rows (or NULL) gathers A rows: logical row m reads A[a_rows[m]]. */
int pk_gemm(int dtype, int a_is_f32, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
            const float* bias, const float* res, int ldr, void* C, int ldc, int out_is_f32, int act,
            const int* a_rows, void* stream);

/* pk_gemm with an explicit main-loop variant (0 = automatic; 1/2 = register-staged 64x64 / 128x128 tiles; 8, 9, 24, 27, 33, 3 = the
 * LDS-DMA ring variants that survived the round-1 sweep; 50 (bf16, round 6) = the 256x256 two-group loop of csrc/gemm_p8.hpp, what `automatic` picks for
 * K >= 2048 with >= 512 such tiles; see csrc/gemm.hip) and the number of physical rows behind a gathered A (bounds of the DMA descriptor).
 * The DMA variants need A in T and ldw >= K rounded up to the k-tile (64 bf16 / 32 f32) with zero padding.
 * C2 (bf16 [M][ldc2], or NULL): a second copy of an f32 result for the next GEMM's LDS-DMA operand (dtype 1 only).
 * ln_s / ln_t ([N] f32, or both NULL): the LayerNorm that precedes this nn.Linear in the reference (attention.py:47,142,
 * cvivit.py:277) folded into it -- A holds the UN-normalised rows x, W holds gamma (.) W, and the epilogue applies
 *   LN(x) W^T = rstd * (x (gamma.W)^T - mean * ln_s) + ln_t,  ln_s[n] = sum_k gamma[k] W[n][k],  ln_t[n] = sum_k beta[k] W[n][k]
 * with mean / rstd (biased variance over K, ln_eps inside the sqrt) taken from the A tiles of the main loop; then bias / act / res.
 * row_off [M] / col_off [N] (int32 element offsets, or both NULL): scattered f32 output, element (m, n) -> C[row_off[m] + col_off[n]]
 * with col_off[n + r] = col_off[n] + r inside every aligned group of 4 columns -- the un-patchify 'b t h w (c pt p1 p2) -> b c (t pt)
 * (h p1) (w p2)' of cvivit.py:326-334 is such a map, so to_pixels writes the video in place (no res / C2 / GEGLU with it).
 * stats_out [M][ceil(N/32)][2] f32 (or NULL): every 32-column chunk of every output row also leaves (sum, sum of squares) of its values
 * -- as stored in C2 when given, else in C -- for the LayerNorm-folded GEMM that reads those rows next; that GEMM passes the buffer as
 * ln_stats [M][ceil(K/32)][2] beside ln_s / ln_t and takes mean / rstd from it instead of from its own main loop (LDS-DMA variants
 * 8 / 24 / 33 / f32 3 only; partials are added in index order, so results do not depend on scheduling).
 * dup_rows (0, or >= M): every output row m of C (and C2) is also written at row m + dup_rows -- the to_out GEMM of the first
 * self-attention block of a classifier-free-guidance batch writes the cond and the null copy of the residual stream at once
 * (phenaki_pytorch.py:149-161: both forwards see the same ids; they differ from the first cross-attention on). */
int pk_gemm_ex(int dtype, int a_is_f32, const void* A, int lda, const void* W, int ldw, int M, int N, int K,
               const float* bias, const float* res, int ldr, void* C, int ldc, int out_is_f32, int act,
               const int* a_rows, int a_nrows, int variant, void* C2, int ldc2, const float* ln_s, const float* ln_t,
               float ln_eps, const int* row_off, const int* col_off, float* stats_out, const float* ln_stats, int dup_rows,
               void* stream);
/* the variant `0 = automatic` resolves to for a shape (host-only helper; used to label kernels in bench.py) */
int pk_gemm_auto_variant(int dtype, int a_is_f32, int M, int N, int K, int lda, int ldw, int a_nrows);

/* y = LayerNorm(x) * gamma (+ beta), eps inside the sqrt, biased variance.  attention.py:29-36 (gamma-only
 * LayerNorm, beta NULL or the zero buffer), attention.py:47 and cvivit.py:277,284 (nn.LayerNorm).
 * out (T if out_kind == 1 else f32) and/or out2 (f32) receive y; raw (T, optional) receives x itself (the
 * un-normalised K/V source of attention.py:140-144).  OUTPUT rows may be remapped: grp > 0:
 * r -> (r / grp) * gstride + goff + r % grp  (the first-frame / rest-frames concat of cvivit.py:549); pb > 0: rows
 * seen as (a, b, c), b < pb, c < pc, land at (a, c, b) ('b t (h w) <-> b (h w) t', cvivit.py:468,472,488,496). */
int pk_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps, void* out, int ldo,
                 int out_kind, float* out2, int ldo2, void* raw, int ldraw, int M, int D, int grp, int gstride,
                 int goff, int pb, int pc, void* stream);

/* cvivit.py:273-285: Rearrange 'b c (t pt) (h p1) (w p2) -> b t h w (c pt p1 p2)' of frames [f0, f0 + nt*pt) of the
 * (B,C,F,H,W) f32 video, fused with nn.LayerNorm(P), P = C*pt*ph*pw; out[(b,t,h,w)][P] is T (out_kind 1) or f32.
 * weight = bias = NULL: the rearranged rows themselves (no LayerNorm). */
int pk_patchify_ln(const float* video, int B, int C, int F, int H, int W, int f0, int nt, int pt, int ph, int pw,
                   const float* weight, const float* bias, float eps, void* out, int ldo, int out_kind, void* stream);

/* cvivit.py:273-285 up to the Linear, bf16 operands, ONE launch for both frame groups (SURVEY.md 2a "K1"): the patch rows are
 * gathered from the (B,C,F,H,W) f32 video inside the GEMM's A-tile producer (no patch matrix in HBM) and nn.LayerNorm(P) is folded in:
 *   out_g[(b,tt,hh,ww)][n] = rstd * (sum_k bf16(x_k - c) * W_g[n][k] - mean' * s_g[n]) + t_g[n]
 * with c a per-patch centre (below), mean' / rstd the f32 statistics of x - c over the P = C*pt*ph*pw patch features (biased variance,
 * eps inside the sqrt), W_g [N][ldw_g] bf16 = gamma (.) W zero-padded along K to 64, s_g[n] = sum_k W_g[n][k], t_g[n] = sum_k beta[k]
 * W[n][k] + bias[n].  Group g covers frames [f0_g, f0_g + nt_g*pt_g) with pt_g frames per patch (ngroups = 1: group 0 only; pass the
 * long-K group first).  pw a power of two in 8..128, P % 192 == 0, video < 4 GiB;
 * c = the mean of the patch's first 32 features.  The caller applies the nn.LayerNorm(dim) that follows. */
int pk_patch_embed(const float* video, int B, int C, int F, int H, int W, int ph, int pw, int N, float eps, int ngroups,
                   const void* W0, int ldw0, const float* s0, const float* t0, float* out0, int f00, int nt0, int pt0,
                   const void* W1, int ldw1, const float* s1, const float* t1, float* out1, int f01, int nt1, int pt1,
                   int ldo, void* stream);

/* The same product as ROW PANELS x ALL COLUMNS x K-SLICES (round 4; cvivit.py:273-285): a workgroup owns 128 patch rows, all N <= 512 output
 * columns and one K-slice of 1024 features -- the f32 video is read exactly once, 204 workgroups at B = 8.  pk_patch_embed_splitk writes, per
 * group, the raw partial products part_g [slices_g][rows_g][N] f32 (sum_k bf16(x_k - c) W_g[n][k] over the slice) and the partial row statistics
 * stats_g [slices_g][rows_g][2] f32 (sum, sum of squares of x - c); slices_g = pk_patch_embed_slices(C*pt_g*ph*pw).  pk_patch_embed_finish adds
 * the slices in index order, applies the folded nn.LayerNorm(P) (s, t = W beta + bias, eps1, K = P) AND the nn.LayerNorm(N) that follows
 * (gamma2, beta2, eps2), writing the token rows: out2 f32 and / or out bf16, row r -> (r / remap_in) * remap_out + remap_off + r % remap_in. */
int pk_patch_embed_slices(int K);
int pk_patch_embed_splitk(const float* video, int B, int C, int F, int H, int W, int ph, int pw, int N, int ngroups,
                          const void* W0, int ldw0, float* part0, float* stats0, int f00, int nt0, int pt0,
                          const void* W1, int ldw1, float* part1, float* stats1, int f01, int nt1, int pt1, void* stream);
int pk_patch_embed_finish(const float* part, const float* stats, int nslices, int rows, int N, int K, const float* s, const float* t,
                          float eps1, const float* gamma2, const float* beta2, float eps2, float* out2, int ldo2, void* out, int ldo,
                          int remap_in, int remap_out, int remap_off, void* stream);
/* pk_patch_embed_finish for BOTH frame groups in one launch (round 5): g[0 .. ngroups) carry the per-group fields above, every group writes the same
 * out2 / out token buffers through its own row remap. */
typedef struct {
    const float* part; const float* stats; int nslices, rows, K;
    const float* s; const float* t; float eps1;
    const float* gamma2; const float* beta2; float eps2;
    int remap_in, remap_out, remap_off;
} pk_patch_finish_group;
int pk_patch_embed_finish_groups(const pk_patch_finish_group* g, int ngroups, int N, float* out2, int ldo2, void* out, int ldo, void* stream);


/* cvivit.py:326-334: Rearrange 'b t h w (c pt p1 p2) -> b c (t pt) (h p1) (w p2)' into frames [f0, f0 + nt*pt). */
int pk_unpatchify(const float* pix, int ldp, float* video, int B, int C, int F, int H, int W, int f0, int nt,
                  int pt, int ph, int pw, void* stream);

/* cvivit.py:585-589 in patch layout: dst[(b,tt,hh,ww)][(c,dt,y,x)] = fmask[b][f0 + tt*pt + dt] ? src[..] : 0 for a (rows, P) f32 matrix
 * (fmask (B, F) uint8; dst may be src): the frame mask of variable-length training where the training step takes its loss (train_cvivit.py). */
int pk_patch_frame_mask(const float* src, long long lds, float* dst, long long ldd, const unsigned char* fmask, int B, int C, int F, int H, int W,
                        int f0, int nt, int pt, int ph, int pw, void* stream);

/* Numerator of the reconstruction loss of cvivit.py:585-591 (F.mse_loss(video, recon), optionally over the frames a
 * (B, F) mask keeps): partials[i], i < PK_SQDIFF_BLOCKS, are per-workgroup double sums of (a - b)^2 over two contiguous
 * (B, C, F, H, W) f32 videos; the caller adds them and divides by the kept element count. */
#define PK_SQDIFF_BLOCKS 1024
int pk_sqdiff_partials(const float* a, const float* b, const unsigned char* fmask, int B, int C, int F, int H, int W,
                       double* partials, void* stream);

/* attention.py:57-85 + residual of attention.py:323: out = x + bias + depthwise_conv3d_3x3x3(zero-padded x) on the
 * channels-last (B,T,H,W,D) reinterpretation of the token buffer; time pad (2,0) if causal else (1,1).
 * wt is dsconv.weight (D,1,3,3,3) pre-transposed to [27][D].  Not in-place. */
int pk_peg(const float* x, const float* wt, const float* bias, float* out, void* out_t, int B, int T, int H, int W, int D,
           int causal, void* stream);       /* out_t: optional bf16 copy of out (the next GEMM's operand) */

/* LFQ (vector-quantize-pytorch, un-vendored; call sites cvivit.py:570 and :439; restated in oracle/lfq.py).
 * encode: proj = x @ Wp^T + bp (f32), ids = sum_k (proj_k > 0) << (cd-1-k); proj (M x cd) optional output.
 * decode: out = (bit ? +1 : -1) @ Wo^T + bo,  Wo is [D][cd].  ids_prime (or NULL): row (b, i) of the M = nb * (n_prime + n) rows takes
 *   its id from ids_prime[b][i] for i < n_prime and ids[b][i - n_prime] otherwise (phenaki_pytorch.py:535-536: the primed tokens are
 *   prepended before decoding -- no concatenated copy); pb, pc > 0: row (a, b, c) is written at (a, c, b), the '(b h w) t' order
 *   the decoder's temporal transformer reads (cvivit.py:482), pb = pc = 0: identity. */
int pk_lfq_encode(const float* x, int ldx, const float* wp, const float* bp, long long* ids, float* proj,
                  int M, int D, int cd, void* stream);
int pk_lfq_decode(const long long* ids, const float* wo, const float* bo, float* out, int M, int D, int cd,
                  const long long* ids_prime, int n_prime, int n, int pb, int pc, void* stream);

/* LFQ training-mode auxiliary loss (third return of `self.vq(tokens)`, cvivit.py:570, added to the GAN generator objective at cvivit.py:666;
 * the published vector-quantize-pytorch LFQ.forward, restated in oracle/lfq.py):
 *   aux = w_e (mean_i H(prob_i) - gamma H(mean_i prob_i)) + w_c mean (z - sign(z) scale)^2,  prob_i = softmax over the 2^cd sign codes of
 *   2 T <z_i, code>,  H(p) = sum -p log(max(p, 1e-5)),  z = proj (M, cd) = project_in(x).
 * The (M, 2^cd) probability matrix is never formed: prob_i(u, v) = a_i[u] b_i[v] over the high / low halves of the bits (hi = ceil(cd / 2),
 * lo = cd / 2; code index = sum_k bit_k 2^(cd-1-k), MSB first as the module's `mask`; alpha = 4 T scale; 2 <= cd <= 16).
 *   pk_lfq_aux_prep:     A (M, 2^hi), B (M, 2^lo) factor rows; ent[i] = H(prob_i); commit[i] = sum_k (z_ik - sign(z_ik) scale)^2
 *   (host: Q = A^T B by pk_bmm = M times the batch distribution)
 *   pk_lfq_aux_codebook: hc_part[pk_lfq_aux_parts(cd)] = partial sums of H(Q inv_n); G = coef * dH/dq at q = Q inv_n  (2^hi x 2^lo)
 *   (host: GA = B G^T, GB = A G by pk_bmm)
 *   pk_lfq_aux_grad:     dproj (M, cd) = d aux / d z  with wen = w_e / M, wc2 = 2 w_c / (M cd), coef above = -w_e gamma / M
 *   pk_lfq_aux_finish:   out[0] = aux, out[1] = mean_i H(prob_i), out[2] = H(batch distribution), out[3] = commitment (fixed summation order). */
int pk_lfq_aux_parts(int cd);
int pk_lfq_aux_prep(const float* proj, int M, int cd, float alpha, float scale, float* A, float* B, float* ent, float* commit, void* stream);
int pk_lfq_aux_codebook(const float* Q, int cd, float inv_n, float coef, float* G, float* hc_part, void* stream);
int pk_lfq_aux_grad(const float* proj, const float* GA, const float* GB, int M, int cd, float alpha, float scale, float wen, float wc2,
                    float* dproj, void* stream);
int pk_lfq_aux_finish(const float* ent, const float* commit, int M, const float* hc_part, int cd, float w_e, float gamma, float w_c, float* out,
                      void* stream);

/* Text-encoder support (reference t5.py:64-103 calls HuggingFace T5EncoderModel; the T5 v1.1 encoder layers are built
 * from pk_gemm, pk_attn_prep with q_scale = k_scale = NULL (plain dot-product attention: no l2norm, q * scale), pk_attn_fwd and these two):
 * pk_rmsnorm: T5LayerNorm, y = x * rsqrt(mean(x^2) + eps) * w (f32 statistics, no mean subtraction, no bias); rows with rowmask[row] == 0
 *   (or NULL: none) are written as zeros (the masked_fill of t5.py:97-100); out is T (out_kind 1) or f32.
 * pk_gated_gelu_tanh: out[m][f] = gelu_new(h[m][f]) * h[m][F + f] for h (M, >= 2F) f32 -- T5DenseGatedActDense's activation on the output
 *   of one GEMM against [wi_0 ; wi_1]. */
int pk_rmsnorm(const float* x, int ldx, const float* w, float eps, const unsigned char* rowmask, void* out, int ldo, int out_kind,
               int M, int D, void* stream);
int pk_gated_gelu_tanh(const float* h, int ldh, void* out, int ldo, int out_kind, int M, int F, void* stream);

/* cvivit.py:472 (the encoder's final norm_out) fused with the LFQ of cvivit.py:570: ids[orow] = LFQ(LayerNorm(x[row]) * gamma
 * (+ beta)) in one pass over x, cd <= 16; orow = the (a, b, c) -> (a, c, b) row permutation of pk_layernorm (pb = 0: identity).
 * tokens (f32 [M][ldt], or NULL) receives the normalised rows, proj ([M][cd], or NULL) the pre-sign projections. */
int pk_layernorm_lfq(const float* x, int ldx, const float* gamma, const float* beta, float eps, const float* wp,
                     const float* bp, long long* ids, float* tokens, int ldt, float* proj, int M, int D, int cd,
                     int pb, int pc, void* stream);

/* rows of x scaled to unit l2 norm (F.normalize, eps 1e-12), out in T (out_kind 1) or f32: the query side of the cosine-sim
 * VectorQuantize lookup (cvivit.py:321). */
int pk_l2norm_rows(const float* x, int ldx, void* out, int ldo, int out_kind, int M, int D, void* stream);

/* phenaki_pytorch.py:194-197, 290-291 (+ the prime-token concat of :500): S sequences of n_tot = n_prime + n positions,
 * out[s*n_tot + i] = token_emb[id] + pos_emb[i] with id = ids_prime[b][i] for i < n_prime (NULL if n_prime == 0), else
 * ids[b][i - n_prime], b = s % nb -- the cond and null replicas of a classifier-free-guidance batch (S = 2 nb) share ids. */
int pk_embed(const long long* ids_prime, int n_prime, const long long* ids, int n, int nb, const float* tok,
             const float* pos, float* out, void* out_t, int S, int D, void* stream);       /* out_t: optional bf16 copy */

/* attention.py:257-272 first CPB layer: out[(i,j)][D] = leaky_relu(W0 @ (sign(rel) log(|rel|+1)) + b0, 0.1) over the
 * flattened (d0,d1,d2) grid; nd = 2 uses (d1,d2) with d0 = 1. */
int pk_cpb_input(const float* w0, const float* b0, float* out, int d0, int d1, int d2, int nd, int D, void* stream);

/* attention.py:146-157: head split, null-kv prepend, l2norm (k after the concat), q/k scales, sim scale folded in q;
 * writes Qp [S][h][nq_pad][64], Kp [S][h][nk_pad][64], Vt [S][h][64][nk_pad] in T. pk_attn_pads gives the pads.
 * kv == NULL prepares the query side only (cross-attention K/V cached across sampling steps). */
int pk_attn_pads(int nq, int n_kv, int nnull, int* nq_pad, int* nk_pad);
int pk_attn_prep(int dtype, const float* q, int ldq, const float* kv, int ldkv, const float* null_kv,
                 const float* q_scale, const float* k_scale, float scale, void* Qp, void* Kp, void* Vt,
                 int S, int h, int nq, int n_kv, int nnull, void* stream);

/* attention.py:142-157 in ONE launch (dtype 1 = bf16: bf16 rows / weights / images; dtype 2 = split-bf16, round 4: f32 rows split in
 * registers, host-split weight planes as for pk_gemm, images written pre-split as for pk_attn_prep(dtype 2)): to_q (A = xq = LayerNorm(x)) and to_kv (A = xkv = the un-normalised x, or NULL
 * for the query side only) as one MFMA GEMM whose epilogue does the head split, l2norm, q/k scales (sim scale folded in
 * q) and the V transpose, writing Qp / Kp / Vt (layouts above) directly: the f32 q / kv matrices never reach HBM and
 * pk_attn_prep is not needed.  Self-attention only on the kv side (no null keys); M = S * nseq rows. */
int pk_qkv_project(int dtype, const void* xq, const void* xkv, int ld, const void* wq, const void* wkv, int ldw, int S, int nseq,
                   int h, int K, const float* q_scale, const float* k_scale, float scale, void* Qp, void* Kp, void* Vt,
                   int nq_pad, int nk_pad, const float* q_ln_s, void* stream);
/* q_ln_s ([h*64] f32, or NULL): the attention's LayerNorm folded into to_q -- xq then holds the un-normalised rows (= xkv), wq holds
 * gamma (.) Wq and q_ln_s its row sums; q = l2norm(x (gamma.Wq)^T - mean(x) q_ln_s) (the l2norm cancels the LayerNorm's rstd). */

/* attention.py:142-182 for SHORT self-attention sequences (n <= 64, no null keys, no key mask; dtype 1 bf16 / 2 split-bf16) in ONE launch: the
 * to_q / to_kv projections, l2norm + scales, and softmax(q k^T + bias (+ ALiBi, causal)) v -- what pk_qkv_project + pk_attn_fwd
 * compute, without Qp / Kp / Vt ever reaching HBM.  A workgroup owns one head of floor(64 / n) whole sequences.  xq = LayerNorm(x)
 * rows, xkv = x rows (both [S*n][ld], bf16 / f32 by dtype); bias [h][n][n] f32 or NULL; slopes [h] with causal; O [S*n][ldo] (bf16 / f32),
 * heads merged.  dtype 2 stages q^ / k^ / v^T as (hi, lo) bf16 planes in LDS and computes every product as three MFMAs. */
int pk_qkv_attn(int dtype, const void* xq, const void* xkv, int ld, const void* wq, const void* wkv, int ldw, int S, int n, int h, int K,
                const float* q_scale, const float* k_scale, float scale, const float* bias, long bias_hstride, int bias_ld,
                const float* slopes, int causal, void* O, int ldo, const float* q_ln_s, void* stream);

/* Cross-attention against CACHED key / value images (the step-invariant text context, attention.py:142-182 with context) in ONE launch:
 * to_q (+ the folded LayerNorm, q_ln_s as in pk_qkv_project) + l2norm + softmax(q k^T (+ key mask)) v.  Kp / Vt are the images
 * pk_attn_prep wrote (same dtype) for nnull + n_kv <= 64 keys (nk_pad from pk_attn_pads); n % 64 == 0; kmask [S][n_kv] uint8 or NULL;
 * O bf16 (dtype 1) / f32 (dtype 2). */
int pk_q_attn_cached(int dtype, const void* xq, int ld, const void* wq, int ldw, int S, int n, int h, int K, const float* q_scale, float scale,
                     const float* q_ln_s, const void* Kp, const void* Vt, int nk_pad, int n_kv, int nnull,
                     const unsigned char* kmask, void* O, int ldo, void* stream);

/* attention.py:157-182: softmax(sim + bias (+ key mask, + ALiBi, causal)) @ v, heads merged: O[(s,i)][hh*64 + d].
 * bias[hh][i][j] is over the real (non-null) keys; kmask [S][n_kv] uint8 (1 = keep); slopes [h] with causal. */
int pk_attn_fwd(int dtype, const void* Qp, const void* Kp, const void* Vt, const float* bias, long bias_hstride,
                int bias_ld, const unsigned char* kmask, const float* slopes, int causal, void* O, int ldo,
                int out_is_f32, int S, int h, int nq, int n_kv, int nnull, const float* bias_tab, int tab_len,
                const int* pos_code, int code_off, int tab_run4, float score_bound, void* stream);
/* The training forward (attention.py:157-182 under autograd): the same product, and lse[(s h + hh) nq + i] = log sum_j exp(score row i) for the
 * backward kernels (pk_attn_bwd with bit 1 of its flag word set skips the pass that recomputed it).  LDS-free kernel; no bias table / score bound. */
int pk_attn_fwd_lse(int dtype, const void* Qp, const void* Kp, const void* Vt, const float* bias, long bias_hstride,
                    int bias_ld, const unsigned char* kmask, const float* slopes, int causal, void* O, int ldo,
                    int out_is_f32, int S, int h, int nq, int n_kv, int nnull, float* lse, void* stream);
/* score_bound: an upper bound of sim + bias over every (head, query, key), or NaN.  q^ and k^ are unit vectors times q_scale / k_scale,
 * so |sim| <= scale * max_d |q_scale_d k_scale_d| and the caller knows the maximum of its bias: with a finite bound (and no key
 * mask, not causal, bf16, >= 64 queries and keys) the softmax numerators are p = 2^(s log2(e) - ceil(bound log2(e))) -- no running
 * maximum, no cross-lane max, no accumulator rescale per key tile (the same softmax: it is shift-invariant; an integer shift of
 * the exponent keeps every mantissa, hence the bf16 rounding of p, independent of the tiling).  NaN: running-max flash loop.
 * tab_run4: 1 when pos_code[4k + r] == pos_code[4k] + r for every k (last grid dimension a multiple of 4): the kernel then fetches the
 * 4 table entries of 4 consecutive keys with two 8-byte LDS reads.
 * bias_tab ([h][tab_len] f32, or NULL; then bias must be NULL): the relative-position form of the continuous position bias
 * (attention.py:229-275), bias[hh][i][j] = bias_tab[hh][pos_code[i] - pos_code[j] + code_off] with pos_code [n] int32 -- staged
 * in LDS by the bf16 kernel for nq = n_kv >= 64 (self-attention, no null keys / mask / causal); other shapes: PK_EINVAL. */

/* attention.py:128-182 for short self-attention sequences (n <= 64, no null keys: the C-ViViT spatial / temporal layers)
 * straight from the projection outputs q [S*n][ldq], kv [S*n][ldkv]: one launch instead of pk_attn_prep + pk_attn_fwd;
 * f32 arithmetic in both precision modes; O f32 (out_kind 0) or bf16 (1). */
int pk_attn_small(const float* q, int ldq, const float* kv, int ldkv, const float* q_scale, const float* k_scale,
                  float scale, const float* bias, long bias_hstride, int bias_ld, const unsigned char* kmask,
                  const float* slopes, int causal, void* O, int ldo, int out_kind, int S, int h, int n, void* stream);

/* phenaki_pytorch.py:155-161 applied to the trunk outputs: e = null + (cond - null) * scale for the non-prime
 * positions; x rows are [cond sequences | null sequences] of n_tot tokens; rows (or NULL) selects output rows
 * (flat b * (n_tot - n_prime) + i). */
int pk_cfg_mix(const float* x, int ldx, int nb, int n_tot, int n_prime, const int* rows, int nrows, float scale,
               int has_null, void* out, int ldo, int out_is_f32, int D, void* stream);

/* phenaki_pytorch.py:213 + 88-93 + 506-509 + 547-550, never materialising logits: per row
 * pred = argmax(logits / max(T,1e-10) + gumbel(U)), optional (max, sum exp) for 1 - softmax[pred].
 * U != NULL: PARITY mode, uniform noise read from U[row][V]; U == NULL: counter-hash noise from `seed` (+ *seed_dev when
 * seed_dev != NULL: a device-resident 64-bit word read at run time, so a captured hipGraph replays with fresh noise).
 * need_lse is a flag word: bit 0 = keep the softmax statistics, bit 1 = no noise at all (plain argmax of A @ W^T + bias:
 * the cosine-sim codebook lookup of the VectorQuantize path, cvivit.py:321,568-570).
 * partials: 5 * pk_vocab_ntiles(V) * M 4-byte words of workspace consumed by pk_vocab_reduce, which writes
 * pred[r], ids[r] = where(mask[r], pred, ids[r]) and scores[r] = where(mask[r], 1 - p, -1e4). */
int pk_vocab_ntiles(int V);
int pk_vocab_sample(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, int M, int V, int D,
                    float temperature, const float* U, const int* rows, unsigned long long seed,
                    const unsigned long long* seed_dev, int need_lse, void* partials, void* stream);
/* PARITY sampling on torch's own device RNG stream (SURVEY.md 7 "phase 2"): the gumbel noise of logical row r, column v is bit for bit
 * what `torch.zeros(rows_total, V, device=...).uniform_(0, 1)[r][v]` holds for a generator at (torch_seed, philox_offset) -- the reference's
 * gumbel_noise(logits), phenaki_pytorch.py:88-93 -- generated in the epilogue (Philox4x32-10 with ATen's element -> (subsequence, counter, word)
 * map, csrc/common.hpp torch_uniform); philox_stride = 256 * min(#CUs * (max threads per CU / 256), ceil(rows_total * V / 256)), the caller
 * advances the generator by 4 * ceil(rows_total * V / (4 * philox_stride)).  A seeded reference run on the same GPU draws the same noise. */
int pk_vocab_sample_philox(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, int M, int V, int D, float temperature,
                           const int* rows, unsigned long long torch_seed, unsigned long long philox_offset, unsigned int philox_stride,
                           int need_lse, void* partials, void* stream);
int pk_vocab_reduce(const void* partials, int M, int V, const int* rows, const unsigned char* mask, long long* ids,
                    long long* pred, float* scores, int need_lse, void* stream);

/* The masked-token cross entropy of phenaki_pytorch.py:640-643 without the logits: after a pk_vocab_sample call with
 * need_lse bit 0 set, loss[m] = logsumexp_v(logits[m][:]) - logits[m][targets[r]] with r = rows ? rows[m] : m, where the
 * target logit is the dot product of row m of A with row targets[r] of W (+ bias) -- same A / W / bias as that call. */
int pk_vocab_ce(int dtype, const void* partials, int M, int V, const void* A, int lda, const void* W, int ldw,
                const float* bias, int D, const long long* targets, const int* rows, float* loss, float* lse_out, void* stream);
/* lse_out ([M] f32 or NULL): logsumexp of every row, kept for the backward pass below.
 *
 * Backward of that cross entropy (first training kernel; phenaki_pytorch.py:640-643 under autograd), one slab of
 * Vs vocabulary columns [v0, v0 + Vs) at a time: logits [M][ldl] f32 holds the slab's recomputed logits (pk_gemm of the same A / W rows
 * / bias); g[m][v] = (exp(logit - lse[m]) - [v0 + v == targets[r]]) * scale (r = rows ? rows[m] : m; scale = upstream gradient / number
 * of rows) is written to g [M][ldg] and, transposed with zeroed pad columns, to gT [Vs][ldgt] (ldgt >= M: the K-padded A operand of the
 * dW GEMM), both f32 or bf16 (out_bf16); db [Vs] (or NULL) receives the column sums of g.  The caller then runs dE += g W_slab and
 * dW_slab = gT E through pk_gemm: the (M, V) logits / probabilities never exist. */
int pk_ce_grad_slab(int out_bf16, const float* logits, int ldl, const float* lse, const long long* targets, const int* rows,
                    int M, int Vs, int v0, float scale, const float* scale_dev, void* g, int ldg, void* gT, int ldgt, float* db, void* stream);
/* scale_dev (or NULL): a device scalar multiplied into scale -- the upstream gradient of the loss, read on the device */

/* phenaki_pytorch.py:488-491: mask = zeros.scatter(1, scores.topk(k).indices, 1).bool(); ids = where(mask, mask_id, ids).
 * rows_out (B*k int32, or NULL) receives the flat positions b*n + i of the masked tokens: only those rows need the vocab
 * head in this step (pk_cfg_mix / pk_vocab_sample / pk_vocab_reduce take it as `rows`).  scores_next (B*n, or NULL; must
 * not alias scores) is filled with -1e4: the critic-less confidence scores where(mask, 1 - p, -1e4) of :547-550, whose
 * masked rows pk_vocab_reduce overwrites afterwards. */
int pk_topk_mask(const float* scores, int B, int n, int k, long long mask_id, unsigned char* mask, long long* ids,
                 int* rows_out, float* scores_next, void* stream);

/* phenaki_pytorch.py:246-263, 523-545: critic head Linear(dim,1) + CFG mix + noise_mult * (u - 0.5), prime dropped.
 * u [nb][n] U[0,1) draws, or NULL: drawn in the kernel from the counter hash of seed (+ *seed_dev, see pk_vocab_sample). */
int pk_critic_head(const float* x, int ldx, const float* w, const float* b, int D, int nb, int n_tot, int n_prime,
                   int has_null, float scale, const float* u, float noise_mult, unsigned long long seed,
                   const unsigned long long* seed_dev, float* out, void* stream);

/* ---- training step, SURVEY.md 8f row 1: backward of the MaskGit / TokenCritic trunk (phenaki_pytorch.py:562-687 under autograd; host side
 * phenaki_pytorch_amd/train.py).  Activations and gradients are f32; the matrix products are pk_gemm calls on operands laid down by pk_pack.
 *
 * pk_pack: f32 matrix -> GEMM operand image.  out[r][k] = (transpose ? src[k][r] : src[r][k]) for r < R, k < K, zero for K <= k < Kp; out has ldo
 *   elements per row; kind 0 f32, 1 bf16, 2 the split-bf16 image (128-byte aligned, ldo % 32 == 0).  dX = dY W takes W^T, dW = dY^T X takes
 *   dY^T as A and X^T as the "W" operand (the contraction index of pk_gemm is the contiguous one). */
int pk_pack(const float* src, long long lds, const int* rows, int R, int K, int transpose, void* out, long long ldo, int Kp, int kind, void* stream);
/* Several pk_pack jobs in ONE launch (round 6; the reference has no counterpart: under autograd each of these is an ATen transpose / cast inside
 * F.linear's backward, attention.py:45-53, :117-119).  A job = one output block, flags = transpose | kind << 1, alignment rules of pk_pack except
 * that `out` may be any 16-byte aligned position inside a 128-byte aligned kind-2 image (a column block of it); tile0 / tiles_x are filled in by
 * the library.  pk_pack_multi: count <= 8 HOST jobs, carried in the kernel arguments (the activation transposes of one backward block).
 * pk_pack_table_prepare + pk_pack_table: any number of jobs -- prepare fills a HOST array and returns the tile count of the launch (< 0: error),
 * the caller keeps a copy in DEVICE memory and replays it every step (the persistent operand images of a Transformer's weights). */
typedef struct PkPackJob {
    const float* src; void* out; long long lds, ldo; int R, K, Kp, flags, tile0, tiles_x;
} PkPackJob;
int pk_pack_multi(const void* jobs, int count, void* stream);
int pk_pack_table_prepare(void* jobs, int count);
int pk_pack_table(const void* dev_table, int count, int tiles, void* stream);
/* rows (or NULL) gathers SOURCE rows: out[r][k] = src[rows[k]][r] (transpose) / src[rows[r]][k].   pk_scatter_rows: dst[rows[m]] = src[m], m < M */
int pk_scatter_rows(const float* src, long long lds, const int* rows, float* dst, long long ldd, int M, int D, void* stream);
/* out[c] (accumulate ? += : =) scale * sum_r src[r][c], two deterministic stages (one launch when M <= 1024 and the rows are float4-addressable,
 * or M <= 256: the per-block partials of the backward kernels); work: pk_colsum_parts(M) * N floats */
int pk_colsum_parts(int M);
int pk_colsum(const float* src, long long ld, int M, int N, float scale, float* out, int accumulate, float* work, void* stream);
/* LayerNorm backward (attention.py:29-36, :47): dx = [add +] rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; pg / pb (pb may be NULL):
 * (pk_ln_bwd_parts(M), D) per-block partials of dgamma = sum dy xhat / dbeta = sum dy, finished by pk_colsum */
int pk_ln_bwd_parts(int M);
int pk_layernorm_bwd(const float* x, long long ldx, const float* gamma, const float* dy, long long lddy, const float* add, long long ldadd,
                     float* dx, long long lddx, float* pg, float* pb, float eps, int M, int D, void* stream);
/* GEGLU on stored pre-activations h (M, >= goff + F): out = h[:, :F] * gelu(h[:, goff : goff + F]) (attention.py:40-43), and its backward */
int pk_geglu(const float* h, long long ldh, int goff, float* out, long long ldo, int M, int F, void* stream);
int pk_geglu_bwd(const float* h, long long ldh, int goff, const float* dout, long long ldd, float* dh, long long lddh, int M, int F, void* stream);
/* dz = dy * (y > 0 ? 1 : slope): LeakyReLU backward from the activation's output (position-bias MLP, attention.py:243-247) */
/* the tokenizer's reconstruction step (cvivit.py:585-591 under autograd; the LFQ's straight-through estimator):
 * pk_scaled_diff: out = (a - b) * scale (* *scale_dev when given) over n floats (n % 4 == 0) -- d/da of (scale / 2) sum (a - b)^2;
 * pk_sign: out = z > 0 ? +value : -value (the code of an LFQ projection, cvivit.py:570 -> vector_quantize_pytorch LFQ.forward). */
int pk_scaled_diff(const float* a, const float* b, float scale, const float* scale_dev, float* out, long long n, void* stream);
int pk_sign(const float* z, float value, float* out, long long n, void* stream);
int pk_mul(const float* a, const float* b, float* out, long long n, void* stream);     /* out = a * b (may alias a), n % 4 == 0 */
int pk_leaky_bwd(const float* y, long long ldy, const float* dy, long long lddy, float* dz, long long lddz, int M, int N, float slope, void* stream);
/* PEG backward (attention.py:57-85 + residual :323): dx = dy + transposed stencil of dy; part (NULL: skip): (pk_peg_wgrad_parts(rows), 27, D)
 * partial tap gradients sum dy[pos] x[pos + tap], finished by pk_colsum over 27 D columns (the conv bias gradient is pk_colsum of dy) */
int pk_peg_adjoint(const float* dy, const float* wt, float* dx, int B, int T, int H, int W, int D, int causal, void* stream);   /* dx only, W in {4, 8, 16} (else PK_EINVAL) */
int pk_peg_wgrad_parts(long long rows);
int pk_peg_bwd(const float* dy, const float* x, const float* wt, float* dx, float* part, int B, int T, int H, int W, int D, int causal, void* stream);
/* token + position embedding backward (phenaki_pytorch.py:194-199, alpha = gradient_shrink_alpha): dpos (n, D) overwritten, dtok (zeroed by
 * the caller) += by f32 atomics */
int pk_embed_bwd(const float* dy, const long long* ids, float alpha, float* dtok, float* dpos, int S, int n, int D, void* stream);
/* position bias (attention.py:257-275): out[h][i][j] = tab[code[i] - code[j] + off][h] and its adjoint (dtab zeroed by the caller, f32 atomics) */
int pk_bias_gather(const float* tab, int ldt, const int* code, int off, float* out, int heads, int n, void* stream);
int pk_bias_scatter(const float* dbias, const int* code, int off, float* dtab, int ldt, int heads, int n, void* stream);
/* out[e] = sum_s src[s * stride + e], e < E (E % 4 == 0) */
int pk_sum_batch(const float* src, long long stride, int S, float* out, long long E, void* stream);
/* count <= 8 short column sums in one launch (the one-launch form of pk_colsum: M <= 8192, N % 4 == 0, 16-byte aligned rows; blk0 filled in by the library) */
typedef struct PkColsumJob { const float* src; float* out; long long ld; int M, N, blk0; float scale; } PkColsumJob;
int pk_colsum_multi(const void* jobs, int count, void* stream);
/* both job lists of a backward block (<= 8 each, either may be empty) in ONE launch */
int pk_reduce_multi(const void* sum_jobs, int nsum, const void* col_jobs, int ncol, void* stream);
/* count <= 8 pk_sum_batch jobs in one launch (E4 = E / 4; blk0 is filled in by the library): the K-slice partials of one block's weight gradients */
typedef struct PkSumJob { const float* src; float* out; long long stride, E4; int S, blk0; } PkSumJob;
int pk_sum_batch_multi(const void* jobs, int count, void* stream);
/* critic head + BCE-with-logits forward and backward in one pass (phenaki_pytorch.py:246-249, :673-676): logits / loss_rows (M) optional;
 * labels NULL: logits only; de = dz w with dz = (sigmoid(z) - y) scale; pw (pk_ln_bwd_parts(M), D) / pb (pk_ln_bwd_parts(M)) partials of dw / db */
int pk_bce_head(const float* e, long long lde, const float* w, const float* b, const float* labels, float scale, const float* scale_dev, float* logits, float* loss_rows,
                float* de, long long ldde, float* pw, float* pb, int M, int D, void* stream);
/* split-K product for the weight gradients (dW = dY^T X contracts over the rows of the batch): C[z] = A[:, z Kc : (z + 1) Kc] W[:, z Kc : (z + 1) Kc]^T,
 * Kc = K / splits (a multiple of the k-tile), z < splits, as `splits` partial (M, N) f32 matrices at C + z * M * ldc; add them with pk_sum_batch.
 * A: T for dtype 1, f32 for dtype 0 / 2; W: the operand image of the dtype.
 * bias ([N] or NULL) is added by slice 0 only (the sum of the slices carries it once); tile 0: 64 x 64 tiles, 1: 128 x 128 (long K-slices of a large
 * (M, N): the patch-embedding product of the exact-f32 / split-bf16 modes, K = 6144), 2 (dtype 1 only, round 6): 256 x 256 tiles on the two-group loop of
 * csrc/gemm_p8.hpp -- few tiles, long K. */
int pk_gemm_splitk(int dtype, const void* A, int lda, const void* W, int ldw, int M, int N, int K, int splits, float* C, int ldc,
                   const float* bias, int tile, void* stream);
/* AdamW / Adam update of one parameter tensor (reference optimizer.py:11-37 hands MaskGit's parameters to torch.optim.AdamW / Adam):
 * m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p = p (1 - lr wd) - lr / (1 - b1^step) * m / (sqrt(v / (1 - b2^step)) + eps); step = 1, 2, ... */
int pk_adamw(float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps, float wd, int step, long long n, void* stream);
/* the same update for `count` tensors sharing hyper-parameters and step in a handful of launches: table = HOST array of count x 5 64-bit
 * words {p, g, m, v, numel} (device pointers of contiguous f32 tensors).  Bit-identical to per-tensor pk_adamw calls. */
int pk_adamw_multi(const long long* table, int count, float lr, float beta1, float beta2, float eps, float wd, int step, void* stream);
/* attention backward (attention.py:132-182).  pk_attn_train_prep: the f32 operands q^ = l2norm(q) q_scale scale -> Qh (S heads, n, 64),
 * k^ = l2norm([null_k ; k]) k_scale -> Kh, [null_v ; v] -> Vh (S heads, nnull + n_kv, 64) from the projection outputs q (S n, ldq), kv (S n_kv, ldkv).
 * pk_attn_bwd: dQh / dKh / dVh from those, the forward output O (f32 or bf16) and dO; bias (heads, n, n_kv) / kmask (S, n_kv) cover the real keys;
 * dS (S heads, n, n_kv; optional) = the score gradient for the position-bias gradient (pk_sum_batch over S); lse / Drow: (S heads n) scratch;
 *   causal (attention.py:166-172, the C-ViViT temporal transformers): ALiBi slopes [heads] over all nnull + n keys and the causal mask;
 *   split_bf16: a flag word -- bit 0: the tile products on the bf16 matrix cores from (hi, lo) splits of the f32 operands (the bf16x3 / bf16 modes),
 *   else exact f32; bit 1: `lse` already holds every row's log-sum-exp (pk_attn_fwd_lse), else it is scratch the first kernel fills.
 *   pq / pk may be the two 64-column halves of ONE (1024, 128) buffer (pk == pq + 64: rows are then 128 floats apart) so that one pk_colsum finishes
 *   both; likewise pg / pb of pk_layernorm_bwd (pb == pg + D: rows 2 D apart).
 * pk_attn_train_prep_bwd: back through l2norm / scales / null keys: dq, dkv, partials pq / pk (1024, 64) of dq_scale / dk_scale, dnull (heads, 2 nnull, 64). */
int pk_attn_train_prep(const float* q, long long ldq, const float* kv, long long ldkv, const float* null_kv, const float* q_scale, const float* k_scale,
                       float scale, float* Qh, float* Kh, float* Vh, int S, int heads, int n, int n_kv, int nnull, void* stream);
int pk_attn_train_prep_bwd(const float* q, long long ldq, const float* kv, long long ldkv, const float* null_kv, const float* q_scale, const float* k_scale,
                           float scale, const float* dQh, const float* dKh, const float* dVh, float* dq, long long lddq, float* dkv, long long lddkv,
                           float* pq, float* pk, float* dnull, int S, int heads, int n, int n_kv, int nnull, void* stream);
int pk_attn_bwd(const float* Qh, const float* Kh, const float* Vh, const void* O, long long ldo, int o_bf16, const float* dO, long long lddo,
                const float* bias, const unsigned char* kmask, const float* slopes, int causal, float* dQh, float* dKh, float* dVh, float* dS,
                float* lse, float* Drow, int S, int heads, int n, int n_kv, int nnull, int split_bf16, void* stream);
/* the same with a workspace of pk_attn_bwd_work(...) floats (0: none needed): with few key tiles (cross-attention on 14 keys: one tile per head) the
 * query tiles of a key tile are dealt to several workgroups whose partial dK^ / dV slabs are added in index order (deterministic). */
int pk_attn_bwd_work(int S, int heads, int n, int n_kv, int nnull);
int pk_attn_bwd_ws(const float* Qh, const float* Kh, const float* Vh, const void* O, long long ldo, int o_bf16, const float* dO, long long lddo,
                   const float* bias, const unsigned char* kmask, const float* slopes, int causal, float* dQh, float* dKh, float* dVh, float* dS,
                   float* lse, float* Drow, int S, int heads, int n, int n_kv, int nnull, int split_bf16, float* work, long long work_floats, void* stream);

/* ---- the tokenizer's adversarial branch: cvivit.py:59-213 (Discriminator), :604-671 (hinge / gradient penalty / adaptive weight).
 * Images are CHANNELS-LAST pixel rows x[(b, y, x)][c] (f32, C % 4 == 0), so every nn.Conv2d (cvivit.py:115-127, 191) is pk_gemm on a patch matrix:
 * pk_im2col: cols[(b, yo, xo)][(ky * kw + kx) * C + c] = x[b][yo stride + ky - pad][xo stride + kx - pad][c], zero outside the image
 *   (3x3 / pad 1: the net convolutions; 1x1 / stride 2: conv_res; 2x2 / stride 2: Rearrange('b c (h p1) (w p2) -> b (c p1 p2) h w') + 1x1 conv);
 * pk_col2im: its adjoint in gather form (input gradient of the convolution; deterministic).
 * pk_nchw_to_rows / pk_rows_to_nchw: (B, C, H, W) <-> rows[(b, y, x)][Cp] with channels C..Cp-1 zero / dropped (mutually adjoint).
 * pk_pick_frames: pick_video_frame (cvivit.py:217-224) img[b] = video[b, :, frame[b]] (place = 0), or the adjoint into a zeroed video (place = 1);
 *   frame: B device int32 indices; an index outside [0, F) reads as a zero frame / is not written (never an out-of-bounds access).
 * pk_bmm: C[z] = op(A[z]) op(B[z]) (+ C[z]) in exact f32 for any shape / leading dimension / batch stride (elements): the fallback product of the
 *   second-order graph of gradient_penalty (cvivit.py:59-73) and of the 64-token attention block inside the discriminator (cvivit.py:166-168). */
int pk_im2col(const float* x, int B, int H, int W, int C, int kh, int kw, int stride, int pad, float* cols, long long ldc, void* stream);
int pk_col2im(const float* cols, long long ldc, int B, int H, int W, int C, int kh, int kw, int stride, int pad, float* dx, void* stream);
int pk_nchw_to_rows(const float* img, int B, int C, int H, int W, int Cp, float* rows, void* stream);
int pk_rows_to_nchw(const float* rows, int B, int C, int H, int W, int Cp, float* img, void* stream);
int pk_pick_frames(float* video, const int* frame, int B, int C, int F, int H, int W, float* img, int place, void* stream);
int pk_bmm(const float* A, long long lda, long long sA, int tA, const float* B, long long ldb, long long sB, int tB, float* C, long long ldc,
           long long sC, int batch, int M, int N, int K, int accumulate, void* stream);
/* Row-wise pieces of the discriminator's attention block WITH their second derivatives (gradient_penalty differentiates the input gradient through the
 * block: cvivit.py:59-73, 166-168; attention.py:29-36, 153-155, 176); one wave per row, contiguous f32 rows.
 * pk_row_softmax   mode 0: out = softmax(a);  1: out = a (b - <a, b>)  (a = y, b = dy: the backward);  2: out = c (b - <a, b>) - b <c, a>  (the gradient
 *                  of <c, backward> with respect to y; with respect to dy it is mode 1 with b = c).
 * pk_row_l2scale   mode 0: o0 = x / |x| * sc  (F.normalize, then q_scale / k_scale);  1: o0 = dx, o1 = per-row contributions to dsc, from dz;
 *                  2: o0 = grad_x, o1 = per-row contributions to grad_sc, o2 = grad_dz of <gx, dx> + <gsc, dsc>.
 * pk_row_ln_bwd2   the same for the gamma-only LayerNorm's backward (pk_layernorm_bwd): u = upstream of dx [M][D], w = upstream of dgamma [D].
 * The per-row contributions are summed over rows by pk_colsum. */
int pk_row_softmax(const float* a, const float* b, const float* c, float* out, long long M, int n, int mode, void* stream);
int pk_row_l2scale(const float* x, const float* sc, const float* dz, const float* gx, const float* gsc, float* o0, float* o1, float* o2,
                   long long M, int d, int mode, void* stream);
int pk_row_ln_bwd2(const float* x, const float* gamma, const float* dy, const float* u, const float* w, float eps, float* grad_x,
                   float* grad_gamma_rows, float* grad_dy, long long M, int D, void* stream);

#ifdef __cplusplus
}
#endif
#endif
