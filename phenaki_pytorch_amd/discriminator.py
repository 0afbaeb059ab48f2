"""The tokenizer's adversarial critic on the MI355X kernels: reference /root/reference/phenaki_pytorch/cvivit.py:101-213
(`DiscriminatorBlock`, `Discriminator`), the GAN losses of :59-99 and the callers at :604-671 (train_gan.py).

Same constructor arguments, module tree and state_dict keys as the reference (`blocks.{i}.conv_res / net.{0,2} / downsample.1`,
`attn_blocks.{i}.*`, `to_logits.{0,3}`): a reference checkpoint's `discr.*` entries load unchanged.  The nn.Conv2d / nn.Linear objects are
parameter containers; the arithmetic is

    image (B, C, H, W)  --pk_nchw_to_rows-->  channels-last pixel rows x[(b, y, x)][c]          (C padded to 8 with zero channels)
    nn.Conv2d k x k      =  pk_im2col (patch matrix, column order (ky, kx, c))  +  pk_gemm against the weight re-ordered to (o, ky, kx, c)
    LeakyReLU(0.1)       =  the GEMM epilogue (act 2)
    Rearrange('b c (h p1) (w p2) -> b (c p1 p2) h w') + 1x1 conv   =  the 2x2 / stride-2 patch matrix against the weight re-ordered to (o, p1, p2, c)
    (x + res) / sqrt(2)  =  folded: both summands' weights and biases carry the 1 / sqrt(2) (LeakyReLU is positively homogeneous) and the sum is
                            the residual input of the second GEMM's epilogue
    Attention(dim)       =  the attention block of train.py (or, under the gradient penalty, its second-order-capable composition below)

Every step is a torch.autograd.Function whose backward is again built from these Functions (`_MM` is closed under differentiation: the gradient
of a matrix product is two matrix products), so `torch.autograd.grad(..., create_graph=True)` -- the gradient penalty of cvivit.py:59-73 --
differentiates the input gradient through the same HIP kernels -- the LayerNorm / l2norm x scale / softmax of the ONE 64-token attention block included:
they carry hand-derived second derivatives (pk_row_softmax / pk_row_l2scale / pk_row_ln_bwd2, formulas in csrc/conv.hip).  What stays on ATen:
re-ordering / scaling of the (small) weight tensors, head split / merge copies, and the scalar loss arithmetic on (B,) logits and gradient norms.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from .attention import Attention, PackedModule, compute_dtype_of, round_up
from .train import _f32, _q, _weight_grad_gemm, attention_train, pack_operand, transposes

LEAK = 0.1            # leaky_relu(p = 0.1), cvivit.py:74-75: the slope pk_gemm's act 2 epilogue applies
CPAD = 8              # image channels are padded to a multiple of 8: the patch matrix rows stay 16-byte aligned, K % 8 == 0


def pair(val):
    ret = (val, val) if not isinstance(val, tuple) else val
    assert len(ret) == 2
    return ret


def cast_tuple(val, l=1):
    return val if isinstance(val, tuple) else (val,) * l


# ------------------------------------------------------------------------------------------------ matrix products, closed under differentiation

def _fast_ok(M, N, K):
    """shapes pk_gemm takes: contraction a multiple of 8 (16-byte rows in every mode), vector epilogue (N % 4), enough rows to fill a tile.  The
    8-channel pixel rows of the first block's 1x1 residual convolution (K = 8 forward, N = 8 in its weight gradient over 131072 rows) qualify."""
    return K % 8 == 0 and K >= 8 and M >= 16 and N >= 8 and N % 4 == 0


def _mm_raw(A, B, tA, tB, dtype):
    """op(A) op(B) for f32 tensors without autograd: 2-D operands of GEMM-friendly shape run on pk_gemm in the compute dtype of the module
    (A W^T, A W, and the row-contracted A^T B with split-K); batched (3-D), single-unit and odd shapes on pk_bmm (exact f32)."""
    dev = A.device
    if A.ndim == 3:
        Z = A.shape[0]
        A, B = A.contiguous(), B.contiguous()
        M, K = (A.shape[2], A.shape[1]) if tA else (A.shape[1], A.shape[2])
        N = B.shape[1] if tB else B.shape[2]
        assert (B.shape[2] if tB else B.shape[1]) == K and B.shape[0] == Z
        out = _f32((Z, M, N), dev)
        return L.bmm(A, B, out, tA, tB, Z, M, N, K, lda=A.stride(1), ldb=B.stride(1), ldc=N, sA=A.stride(0), sB=B.stride(0), sC=M * N)
    A, B = A.contiguous(), B.contiguous()
    M, K = (A.shape[1], A.shape[0]) if tA else A.shape
    N = B.shape[0] if tB else B.shape[1]
    assert (B.shape[1] if tB else B.shape[0]) == K, (tuple(A.shape), tuple(B.shape), tA, tB)
    out = _f32((M, N), dev)
    if _fast_ok(M, N, K):
        if not tA:
            # A (M, K) rows against the operand image of op(B)^T = (N, K): B itself when tB, its transpose otherwise
            return L.gemm(dtype, A, pack_operand(B, dtype, transpose=not tB), M, N, K, C=out)
        if not tB and K >= 4 * _q(dtype) and M % 4 == 0:
            # A^T B, contraction over the K rows both operands share: the weight-gradient shape (split-K, added in index order)
            Kp = round_up(K, _q(dtype))
            AT, BT = transposes(dtype, [(A, 'a'), (B, 'w')])               # (M, Kp), (N, Kp): one launch for both
            _weight_grad_gemm(dtype, AT, BT, M, N, Kp, out)
            return out
    return L.bmm(A, B, out, tA, tB, 1, M, N, K, lda=A.stride(0), ldb=B.stride(0), ldc=N)


class _MM(torch.autograd.Function):
    """C = op(A) op(B) (op = transpose when the flag is set).  backward is two more _MM nodes, so any order of derivative exists."""

    @staticmethod
    def forward(ctx, A, B, tA, tB, dtype):
        ctx.save_for_backward(A, B)
        ctx.cfg = (tA, tB, dtype)
        return _mm_raw(A.detach(), B.detach(), tA, tB, dtype)

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        tA, tB, dt = ctx.cfg
        dA = dB = None
        if ctx.needs_input_grad[0]:
            if not tA:
                dA = _MM.apply(dC, B, False, not tB, dt)                   # dC op(B)^T
            else:
                dA = _MM.apply(B, dC, tB, True, dt)                        # op(B) dC^T
        if ctx.needs_input_grad[1]:
            if not tB:
                dB = _MM.apply(A, dC, not tA, False, dt)                   # op(A)^T dC
            else:
                dB = _MM.apply(dC, A, True, tA, dt)                        # dC^T op(A)
        return dA, dB, None, None, None


def mm(A, B, tA=False, tB=False, dtype=L.F32):
    return _MM.apply(A, B, tA, tB, dtype)


class _SlopeMul(torch.autograd.Function):
    """g * (y > 0 ? 1 : 0.1): LeakyReLU's derivative applied to g (y = the activation's output or input: same sign).  Linear in g, constant in y."""

    @staticmethod
    def forward(ctx, g, y):
        ctx.save_for_backward(y)
        g = g.contiguous()
        M, N = y.shape
        out = _f32((M, N), y.device)
        L.leaky_bwd(y, g, out, M, N, LEAK)
        return out

    @staticmethod
    def backward(ctx, gg):
        y, = ctx.saved_tensors
        return _SlopeMul.apply(gg, y), None


class _ColSum(torch.autograd.Function):
    """column sums of a 2-D tensor (the bias gradient); its adjoint is the row broadcast"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.M = x.shape[0]
        return L.colsum(x, x.shape[0], x.shape[1], _f32((x.shape[1],), x.device))

    @staticmethod
    def backward(ctx, g):
        return g.unsqueeze(0).expand(ctx.M, -1)


class _Affine(torch.autograd.Function):
    """y = act(A W^T + bias) (+ res) in ONE pk_gemm launch (act: LeakyReLU(0.1) in the epilogue; res: the residual input of the epilogue).
    A (M, K) f32 rows with K % 8 == 0 -- a patch matrix or pixel rows --, W (N, K)."""

    @staticmethod
    def forward(ctx, A, W, bias, res, act, dtype):
        M, K = A.shape
        N = W.shape[0]
        y = _f32((M, N), A.device)
        L.gemm(dtype, A.detach().contiguous(), pack_operand(W.detach().contiguous(), dtype), M, N, K, C=y,
               bias=bias.detach().contiguous() if bias is not None else None,
               res=res.detach().contiguous() if res is not None else None, act=L.ACT_LEAKY if act else L.ACT_NONE)
        ctx.cfg = (act, dtype)
        ctx.has_res = res is not None
        if act and res is not None:
            raise AssertionError('the epilogue adds the residual after the activation: the mask would need the pre-residual value')
        ctx.save_for_backward(A, W, y if act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        A, W, y = ctx.saved_tensors
        act, dt = ctx.cfg
        dz = _SlopeMul.apply(dy, y) if act else dy
        dA = mm(dz, W, False, False, dt) if ctx.needs_input_grad[0] else None
        dW = mm(dz, A, True, False, dt) if ctx.needs_input_grad[1] else None
        db = _ColSum.apply(dz) if ctx.needs_input_grad[2] else None
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dA, dW, db, dres, None, None


# ------------------------------------------------------------------------------------------------ layout maps and their adjoints

def _geom_out(geom):
    B, H, W, C, kh, kw, stride, pad = geom
    return L.conv_out_size(H, kh, stride, pad), L.conv_out_size(W, kw, stride, pad)


class _Im2col(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geom):
        B, H, W, C, kh, kw, stride, pad = geom
        Ho, Wo = _geom_out(geom)
        ctx.geom = geom
        return L.im2col(x.contiguous(), B, H, W, C, kh, kw, stride, pad, _f32((B * Ho * Wo, kh * kw * C), x.device))

    @staticmethod
    def backward(ctx, d):
        return _Col2im.apply(d, ctx.geom), None


class _Col2im(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cols, geom):
        B, H, W, C, kh, kw, stride, pad = geom
        ctx.geom = geom
        return L.col2im(cols.contiguous(), B, H, W, C, kh, kw, stride, pad, _f32((B * H * W, C), cols.device))

    @staticmethod
    def backward(ctx, d):
        return _Im2col.apply(d, ctx.geom), None


class _ToRows(torch.autograd.Function):
    """(B, C, H, W) -> pixel rows (B H W, Cp), zero padding channels"""

    @staticmethod
    def forward(ctx, img, Cp):
        ctx.shape, ctx.Cp = tuple(img.shape), Cp
        B, C, H, W = img.shape
        return L.nchw_to_rows(img.contiguous(), Cp, _f32((B * H * W, Cp), img.device))

    @staticmethod
    def backward(ctx, d):
        return _FromRows.apply(d, ctx.shape, ctx.Cp), None


class _FromRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, shape, Cp):
        ctx.Cp = Cp
        return L.rows_to_nchw(rows.contiguous(), Cp, _f32(shape, rows.device))

    @staticmethod
    def backward(ctx, d):
        return _ToRows.apply(d, ctx.Cp), None, None


class PickFrame(torch.autograd.Function):
    """pick_video_frame (cvivit.py:217-224): (B, C, F, H, W), frame (B,) int32 -> (B, C, H, W); adjoint: the frame placed into a zero video"""

    @staticmethod
    def forward(ctx, video, frame):
        ctx.save_for_backward(frame)
        ctx.shape = tuple(video.shape)
        B, C, Fr, H, W = video.shape
        return L.pick_frames(video.contiguous(), frame, _f32((B, C, H, W), video.device))

    @staticmethod
    def backward(ctx, d):
        frame, = ctx.saved_tensors
        return _PlaceFrame.apply(d, frame, ctx.shape), None


class _PlaceFrame(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, frame, shape):
        ctx.save_for_backward(frame)
        video = torch.zeros(shape, device=img.device, dtype=torch.float32)
        return L.pick_frames(video, frame, img.contiguous(), place=True)

    @staticmethod
    def backward(ctx, d):
        frame, = ctx.saved_tensors
        return PickFrame.apply(d, frame), None, None


# ------------------------------------------------------------------------------------------------ modules (parameter containers)

class _Layout(nn.Identity):
    """keeps the reference's nn.Sequential indices where it has a parameter-free einops Rearrange"""


class DiscriminatorBlock(nn.Module):
    """cvivit.py:103-134"""

    def __init__(self, input_channels, filters, downsample=True):
        super().__init__()
        self.conv_res = nn.Conv2d(input_channels, filters, 1, stride=(2 if downsample else 1))
        self.net = nn.Sequential(nn.Conv2d(input_channels, filters, 3, padding=1), nn.LeakyReLU(LEAK),
                                 nn.Conv2d(filters, filters, 3, padding=1), nn.LeakyReLU(LEAK))
        self.downsample = nn.Sequential(_Layout(), nn.Conv2d(filters * 4, filters, 1)) if downsample else None


def _conv_matrix(w, Cp, scale=None):
    """nn.Conv2d weight (O, Cin, kh, kw) -> (O, kh kw Cp) in the patch matrix's column order (ky, kx, c), zero columns for the padding channels"""
    O, Cin, kh, kw = w.shape
    m = w.permute(0, 2, 3, 1)
    if Cp != Cin:
        m = F.pad(m, (0, Cp - Cin))
    m = m.reshape(O, kh * kw * Cp)
    return m * scale if scale is not None else m.contiguous()


def _block_forward(block, x, B, H, W, C, dt):
    """DiscriminatorBlock.forward (cvivit.py:129-138) on pixel rows: returns (rows, H', W', filters)"""
    c = 1.0 / math.sqrt(2.0)
    conv0, conv1 = block.net[0], block.net[2]
    Co = conv0.out_channels
    assert C % 8 == 0 and Co % 8 == 0, 'channel counts must be multiples of 8 (discr_base_dim even)'
    h = _Affine.apply(_Im2col.apply(x, (B, H, W, C, 3, 3, 1, 1)), _conv_matrix(conv0.weight, C), conv0.bias, None, True, dt)
    if block.downsample is not None:
        assert H % 2 == 0 and W % 2 == 0
        h = _Affine.apply(_Im2col.apply(h, (B, H, W, Co, 3, 3, 1, 1)), _conv_matrix(conv1.weight, Co), conv1.bias, None, True, dt)
        res = _Affine.apply(_Im2col.apply(x, (B, H, W, C, 1, 1, 2, 0)), _conv_matrix(block.conv_res.weight, C, c), block.conv_res.bias * c, None, False, dt)
        down = block.downsample[1]
        wd = down.weight.reshape(Co, Co, 2, 2).permute(0, 2, 3, 1).reshape(Co, 4 * Co) * c       # (o, c p1 p2) -> (o, p1 p2 c)
        out = _Affine.apply(_Im2col.apply(h, (B, H, W, Co, 2, 2, 2, 0)), wd, down.bias * c, res, False, dt)
        return out, H // 2, W // 2, Co
    # no down-sampling: leaky(c z) = c leaky(z), so the second convolution carries the 1 / sqrt(2) and the residual convolution adds onto it
    h = _Affine.apply(_Im2col.apply(h, (B, H, W, Co, 3, 3, 1, 1)), _conv_matrix(conv1.weight, Co, c), conv1.bias * c, None, True, dt)
    out = _Affine.apply(x, _conv_matrix(block.conv_res.weight, C, c), block.conv_res.bias * c, h, False, dt)
    return out, H, W, Co


class _Softmax(torch.autograd.Function):
    """softmax over the last dimension; backward = _SoftmaxBwd (itself differentiable once more)"""

    @staticmethod
    def forward(ctx, x):
        y = L.row_softmax(x.contiguous(), None, None, torch.empty_like(x, memory_format=torch.contiguous_format), 0)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        return _SoftmaxBwd.apply(y, dy)


class _SoftmaxBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, dy):
        y, dy = y.contiguous(), dy.contiguous()
        ctx.save_for_backward(y, dy)
        return L.row_softmax(y, dy, None, torch.empty_like(y), 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        y, dy = ctx.saved_tensors
        g = g.contiguous()
        return L.row_softmax(y, dy, g, torch.empty_like(y), 2), L.row_softmax(y, g, None, torch.empty_like(y), 1)


class _L2Scale(torch.autograd.Function):
    """F.normalize(x, dim=-1) * sc (attention.py:153-155: l2norm, then q_scale / k_scale)"""

    @staticmethod
    def forward(ctx, x, sc):
        x, sc = x.contiguous(), sc.contiguous()
        z = torch.empty_like(x)
        L.row_l2scale(x, sc, None, None, None, z, None, None, 0)
        ctx.save_for_backward(x, sc)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, sc = ctx.saved_tensors
        return _L2ScaleBwd.apply(x, sc, dz)


class _L2ScaleBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sc, dz):
        x, sc, dz = x.contiguous(), sc.contiguous(), dz.contiguous()
        d = x.shape[-1]
        dx, rows = torch.empty_like(x), torch.empty_like(x)
        L.row_l2scale(x, sc, dz, None, None, dx, rows, None, 1)
        ctx.save_for_backward(x, sc, dz)
        return dx, L.colsum(rows.view(-1, d), rows.numel() // d, d, _f32((d,), x.device))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gx, gsc):
        x, sc, dz = ctx.saved_tensors
        d = x.shape[-1]
        gx = torch.zeros_like(x) if gx is None else gx.contiguous()
        gsc = torch.zeros_like(sc) if gsc is None else gsc.contiguous()
        grad_x, rows, grad_dz = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        L.row_l2scale(x, sc, dz, gx, gsc, grad_x, rows, grad_dz, 2)
        return grad_x, L.colsum(rows.view(-1, d), rows.numel() // d, d, _f32((d,), x.device)), grad_dz


class _GammaLayerNorm(torch.autograd.Function):
    """attention.py:29-36 (gamma parameter, zero beta buffer) on (M, D) rows: pk_layernorm / pk_layernorm_bwd / pk_row_ln_bwd2"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        M, D = x.shape
        y = _f32((M, D), x.device)
        L.layernorm(x, gamma.detach(), beta, M, D, out2=y, eps=eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg = _GammaLayerNormBwd.apply(x, gamma, dy, ctx.eps)
        return dx, dg, None, None


class _GammaLayerNormBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, dy, eps):
        x, dy = x.contiguous(), dy.contiguous()
        M, D = x.shape
        dx = _f32((M, D), x.device)
        dg, _ = L.layernorm_bwd(x, gamma.detach().contiguous(), dy, dx, M, D, eps=eps)
        ctx.save_for_backward(x, gamma, dy)
        ctx.eps = eps
        return dx, dg

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u, w):
        x, gamma, dy = ctx.saved_tensors
        M, D = x.shape
        u = torch.zeros_like(x) if u is None else u.contiguous()
        w = torch.zeros_like(gamma) if w is None else w.contiguous()
        grad_x, rows, grad_dy = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        L.row_ln_bwd2(x, gamma.detach().contiguous(), dy, u, w, ctx.eps, grad_x, rows, grad_dy)
        return grad_x, L.colsum(rows, M, D, _f32((D,), x.device)), grad_dy, None


def _attention_second_order(attn, x, S, n, dt):
    """attn(x) + x (cvivit.py:166-168; attention.py:132-182 with num_null_kv = 0, no mask / bias) as a graph of Functions that can be differentiated
    twice: LayerNorm, l2norm x scale and softmax carry their own second derivatives (pk_row_*), the five products are _MM / _Affine"""
    assert attn.num_null_kv == 0 and not attn.causal
    h = attn.heads
    xn = _GammaLayerNorm.apply(x, attn.norm.gamma, attn.norm.beta, attn.norm.eps)
    q = mm(xn, attn.to_q.weight, False, True, dt)
    kv = mm(x, attn.to_kv.weight, False, True, dt)                       # K / V from the un-normalised rows (attention.py:140-144)
    k, v = kv.chunk(2, dim=-1)

    def heads(t):
        return t.reshape(S, n, h, 64).permute(0, 2, 1, 3).reshape(S * h, n, 64)
    q, k, v = heads(q), heads(k), heads(v)
    q = _L2Scale.apply(q, attn.q_scale * float(attn.scale))              # the similarity scale rides on q's scale vector
    k = _L2Scale.apply(k, attn.k_scale)
    p = _Softmax.apply(mm(q, k, False, True, L.F32))
    o = mm(p, v, False, False, L.F32)
    o = o.reshape(S, h, n, 64).permute(0, 2, 1, 3).reshape(S * n, h * 64)
    return _Affine.apply(o, attn.to_out.weight, None, x, False, dt)


class Discriminator(PackedModule):
    """cvivit.py:141-213.  forward(x): (B, C, H, W) images -> (B,) logits; `second_order=True` builds the graph the gradient penalty needs."""

    def __init__(self, *, dim, image_size, channels=3, attn_res_layers=(16,), max_dim=512):
        super().__init__()
        image_size = pair(image_size)
        min_image_resolution = min(image_size)
        num_layers = int(math.log2(min_image_resolution) - 2)
        attn_res_layers = cast_tuple(attn_res_layers, num_layers)
        layer_dims = [channels] + [(dim * 4) * (2 ** i) for i in range(num_layers + 1)]
        layer_dims = [min(layer_dim, max_dim) for layer_dim in layer_dims]
        layer_dims_in_out = tuple(zip(layer_dims[:-1], layer_dims[1:]))
        blocks, attn_blocks = [], []
        image_resolution = min_image_resolution
        for ind, (in_chan, out_chan) in enumerate(layer_dims_in_out):
            is_not_last = ind != (len(layer_dims_in_out) - 1)
            blocks.append(DiscriminatorBlock(in_chan, out_chan, downsample=is_not_last))
            attn_blocks.append(Attention(dim=out_chan) if image_resolution in attn_res_layers else None)
            image_resolution //= 2
        self.blocks = nn.ModuleList(blocks)
        self.attn_blocks = nn.ModuleList(attn_blocks)
        dim_last = layer_dims[-1]
        downsample_factor = 2 ** num_layers
        last_fmap_size = tuple(map(lambda n: n // downsample_factor, image_size))
        latent_dim = last_fmap_size[0] * last_fmap_size[1] * dim_last
        self.image_size, self.channels = image_size, channels
        self.to_logits = nn.Sequential(nn.Conv2d(dim_last, dim_last, 3, padding=1), nn.LeakyReLU(LEAK), _Layout(),
                                       nn.Linear(latent_dim, 1), _Layout())

    def forward(self, x, second_order=False):
        L.require_device(x, 'images')
        assert x.ndim == 4 and x.shape[1] == self.channels
        dt = compute_dtype_of(self)
        B, C, H, W = x.shape
        Cp = round_up(C, CPAD)
        rows = _ToRows.apply(x.float(), Cp)
        C = Cp
        for block, attn in zip(self.blocks, self.attn_blocks):
            rows, H, W, C = _block_forward(block, rows, B, H, W, C, dt)
            if attn is not None:
                if second_order:
                    rows = _attention_second_order(attn, rows, B, H * W, dt)
                else:
                    rows = attention_train(attn, rows, B, H * W, dt)           # x + to_out(attention(...)): the residual is inside
        conv, lin = self.to_logits[0], self.to_logits[3]
        feat = _Affine.apply(_Im2col.apply(rows, (B, H, W, C, 3, 3, 1, 1)), _conv_matrix(conv.weight, C), conv.bias, None, True, dt)
        assert lin.in_features == H * W * C, 'image size does not match the discriminator it was built for'
        w = lin.weight.reshape(1, C, H, W).permute(0, 2, 3, 1).reshape(1, H * W * C)     # 'b c h w -> b (c h w)' against rows in (h, w, c) order
        logits = mm(feat.reshape(B, H * W * C), w, False, True, dt).reshape(B)
        return logits + lin.bias


# ------------------------------------------------------------------------------------------------ GAN losses (cvivit.py:59-99)

def gradient_penalty(images, output, weight=10):
    """cvivit.py:59-73: weight * mean_b (|| d sum(output) / d images_b ||_2 - 1)^2, differentiable w.r.t. the discriminator's parameters"""
    gradients, = torch.autograd.grad(outputs=output, inputs=images, grad_outputs=torch.ones_like(output), create_graph=True,
                                     retain_graph=True, only_inputs=True)
    gradients = gradients.reshape(images.shape[0], -1)
    return weight * ((gradients.norm(2, dim=1) - 1) ** 2).mean()


def hinge_discr_loss(fake, real):
    return (F.relu(1 + fake) + F.relu(1 - real)).mean()


def hinge_gen_loss(fake):
    return -fake.mean()


def _log(t, eps=1e-10):
    return torch.log(t + eps)


def bce_discr_loss(fake, real):
    """cvivit.py:91-92 (the reference calls an undefined `log` there -- NameError; this is the evident intent, log(t + 1e-10) as in
    phenaki_pytorch.py:59-60)"""
    return (-_log(1 - torch.sigmoid(fake)) - _log(torch.sigmoid(real))).mean()


def bce_gen_loss(fake):
    return -_log(torch.sigmoid(fake)).mean()
