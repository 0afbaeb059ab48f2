"""First training kernel of the MI355X build: the masked-token cross entropy of the MaskGit vocabulary head,
forward AND backward, without ever writing the (rows, 65 536) logits -- reference `F.cross_entropy(logits[mask], ids[mask])` under
autograd, /root/reference/phenaki_pytorch/phenaki_pytorch.py:640-643 (caller phenaki_trainer.py:351-388).

    loss = vocab_cross_entropy(embeds, weight, bias, targets)      # embeds (M, D) = the trunk output rows of the masked positions
    loss.backward()                                                # fills embeds.grad, weight.grad, bias.grad

forward : pk_vocab_sample (no noise, statistics only) + pk_vocab_ce -> per-row lse and loss; nothing of size M x V is stored.
backward: dlogit = (softmax - onehot) / M.  The vocabulary is walked in slabs of `slab` columns: pk_gemm recomputes the slab's
          logits, pk_ce_grad_slab turns them into g (and g^T) in place, and two more pk_gemm calls accumulate dE += g W_slab and write
          dW_slab = g^T E; db_slab = column sums of g.  8 M V D flops in total (recompute + two gradient products); the transient
          slab buffers (3 x M x slab) stay in the Infinity Cache.
The rest of Phenaki.forward (trunk backward, critic BCE) has no backward kernels yet: `Phenaki.forward` still returns a value whose
`.backward()` raises.  Compute dtype: 'fp32' | 'bf16x3' (1e-3-grade gradients) | 'bf16'.
"""
import torch

from . import _lib as L
from .attention import pack_linear_weight, resolve_dtype, round_up


def _operand(x, dtype):
    """(rows, K) f32 -> the `W`-side GEMM operand image of compute dtype `dtype` (K zero-padded to the k-tile)"""
    return pack_linear_weight(x, dtype)


class _VocabCrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embeds, weight, bias, targets, dtype, slab):
        L.require_device(embeds, 'embeds')
        M, D = embeds.shape
        V = weight.shape[0]
        dev = embeds.device
        E = embeds.detach().float().contiguous()
        td = L.tdtype(dtype)
        A = E.to(td) if td != torch.float32 else E
        Wp = _operand(weight.detach().float(), dtype)
        b = bias.detach().float().contiguous() if bias is not None else torch.zeros((V,), device=dev)
        tg = targets.detach().long().contiguous()
        partials = torch.empty((5 * L.vocab_ntiles(V) * M,), device=dev, dtype=torch.float32)
        L.vocab_sample(dtype, A, Wp, b, M, V, D, 1.0, None, None, 0, True, partials, no_noise=True)
        loss_rows = torch.empty((M,), device=dev, dtype=torch.float32)
        lse = torch.empty((M,), device=dev, dtype=torch.float32)
        L.vocab_ce(dtype, partials, M, V, A, Wp, b, D, tg, None, loss_rows, lse=lse)
        ctx.save_for_backward(E, weight, b, tg, lse)
        ctx.dtype, ctx.slab, ctx.has_bias = dtype, slab, bias is not None
        ctx.A, ctx.Wp = A, Wp
        return loss_rows.mean()

    @staticmethod
    def backward(ctx, grad_out):
        E, weight, b, tg, lse = ctx.saved_tensors
        dtype, slab = ctx.dtype, ctx.slab
        A, Wp = ctx.A, ctx.Wp
        M, D = E.shape
        V = weight.shape[0]
        dev = E.device
        td = L.tdtype(dtype)
        q = 64 if dtype == L.BF16 else 32
        Mp, Vp = round_up(M, q), round_up(V, q)
        scale = float(grad_out) / M
        # operands of the two gradient products (W-side images: rows = output features, K along the contraction)
        Wt = _operand(weight.detach().float().t().contiguous(), dtype)         # (D, Vp): dE = g @ W   -> "W" operand = W^T, K = vocabulary
        Et = _operand(E.t().contiguous(), dtype)                               # (D, Mp): dW = g^T @ E -> "W" operand = E^T, K = rows
        dE = torch.empty((M, D), device=dev, dtype=torch.float32)
        dW = torch.empty((V, D), device=dev, dtype=torch.float32)
        db = torch.empty((V,), device=dev, dtype=torch.float32)
        Vs_max = min(slab, V)
        logits = torch.empty((M, Vs_max), device=dev, dtype=torch.float32)
        g = torch.empty((M, Vs_max), device=dev, dtype=td)
        gT = torch.empty((Vs_max, Mp), device=dev, dtype=td)
        first = True
        for v0 in range(0, V, slab):
            Vs = min(slab, V - v0)
            # logits of the slab: the same A / W rows / bias as the forward pass (rows [v0, v0 + Vs) of the packed weight)
            L.gemm(dtype, A, Wp[v0:v0 + Vs], M, Vs, D, C=logits, bias=b[v0:v0 + Vs], ldc=logits.stride(0))
            L.ce_grad_slab(logits, lse, tg, None, M, Vs, v0, scale, g, gT, db=db[v0:v0 + Vs])
            # dE (+)= g @ W_slab: contraction over the slab's columns = columns [v0, v0 + Vs) of the W^T image (k offset on the operand)
            L.gemm(dtype, g, _k_slice(Wt, v0, dtype), M, D, Vs, C=dE, res=None if first else dE, lda=g.stride(0))
            # dW_slab = g^T @ E (K = the rows, padded to the k-tile: gT's pad columns are zeroed by the kernel, E^T is zero-padded)
            L.gemm(dtype, gT, Et, Vs, D, Mp, C=dW[v0:v0 + Vs], lda=gT.stride(0))
            first = False
        return dE.to(ctx.saved_tensors[0].dtype), dW, (db if ctx.has_bias else None), None, None, None


def _k_slice(Wimg, k0, dtype):
    """the operand image `Wimg` (N, Kpad) viewed from contraction index k0 on (k0 a multiple of the k-tile): a column-offset view with
    the same row stride.  Valid for all three layouts -- plain f32 / bf16 rows, and the split-bf16 image whose 32-element blocks keep
    4 bytes per element."""
    return Wimg[:, k0:]


def vocab_cross_entropy(embeds, weight, bias, targets, compute_dtype='bf16x3', slab=2048):
    """mean_m CE(embeds[m] @ weight^T + bias, targets[m]) with gradients for embeds / weight / bias; logits never stored.
    embeds (M, D) f32 on the HIP device, weight (V, D), bias (V,) or None, targets (M,) int64 in [0, V).  slab: vocabulary columns per
    backward step (a multiple of 64)."""
    assert slab % 64 == 0 and slab > 0
    assert weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0, 'vocabulary size and embedding width must be multiples of 8'
    return _VocabCrossEntropy.apply(embeds, weight, bias, targets, resolve_dtype(compute_dtype), int(slab))
