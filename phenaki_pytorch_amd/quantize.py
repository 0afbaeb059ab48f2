"""Lookup-free quantizer with the module surface the reference expects from `vector_quantize_pytorch.LFQ`
(call sites /root/reference/phenaki_pytorch/cvivit.py:319, :570, :439; the package itself is not vendored --
behaviour restated from its published semantics, see oracle/lfq.py and SURVEY.md 8c).

state_dict keys: project_in.{weight (cd, dim), bias}, project_out.{weight (dim, cd), bias}, buffer `mask`.
Always exact f32: the ids are 16 sign bits, so this projection never runs in bf16.
"""
import math

import torch
from torch import nn

from . import _lib as L


class LFQ(nn.Module):
    def __init__(self, *, dim, codebook_size, codebook_scale=1.0, entropy_loss_weight=0.1, commitment_loss_weight=0.25, diversity_gamma=1.,
                 **_unused):
        """the keyword arguments are the published module's (they arrive through CViViT(lookup_free_quantization_kwargs=...), cvivit.py:319)"""
        super().__init__()
        self.entropy_loss_weight, self.commitment_loss_weight, self.diversity_gamma = entropy_loss_weight, commitment_loss_weight, diversity_gamma
        cd = int(math.log2(codebook_size))
        assert 2 ** cd == codebook_size, 'codebook size must be a power of two'
        assert dim != cd, 'the MI355X build expects dim != log2(codebook_size) (projections present)'
        assert codebook_scale == 1.0, 'only codebook_scale = 1 is built'
        self.dim, self.codebook_dim, self.codebook_size = dim, cd, codebook_size
        self.codebook_scale = codebook_scale
        self.project_in = nn.Linear(dim, cd)
        self.project_out = nn.Linear(cd, dim)
        self.register_buffer('mask', 2 ** torch.arange(cd - 1, -1, -1))

    def encode_ids(self, x2d, return_proj=False):
        """x2d (M, dim) f32 -> ids (M,) int64 [, proj (M, cd) f32]"""
        L.require_device(x2d, 'tokens')
        M, D = x2d.shape
        ids = torch.empty((M,), device=x2d.device, dtype=torch.int64)
        proj = torch.empty((M, self.codebook_dim), device=x2d.device, dtype=torch.float32) if return_proj else None
        L.lfq_encode(x2d, self.project_in.weight, self.project_in.bias, ids, proj, M, D, self.codebook_dim)
        return (ids, proj) if return_proj else ids

    def encode_ids_from_prenorm(self, x2d, norm, perm=(0, 0), return_proj=False):
        """ids of LayerNorm(x2d) with the (gamma, beta) of `norm` (the encoder's norm_out), rows permuted like pk_layernorm's
        (pb, pc): ONE fused launch, the normalised tokens never reach HBM (cvivit.py:472 + :570).  Needs cd <= 16."""
        L.require_device(x2d, 'tokens')
        M, D = x2d.shape
        ids = torch.empty((M,), device=x2d.device, dtype=torch.int64)
        proj = torch.empty((M, self.codebook_dim), device=x2d.device, dtype=torch.float32) if return_proj else None
        L.layernorm_lfq(x2d, norm.gamma, norm.beta, self.project_in.weight, self.project_in.bias, ids, M, D, self.codebook_dim,
                        proj=proj, perm=perm)
        return (ids, proj) if return_proj else ids

    def codes_2d(self, ids_flat, *, ids_prime=None, perm=(0, 0)):
        """ids (M,) int64 -> project_out(+-1 codes) (M, dim) f32.  ids_prime (nb, n_prime): primed tokens in front of every sequence of
        `ids_flat` given as (nb, n); perm = (pb, pc): output rows (a, b, c) -> (a, c, b) (see pk_lfq_decode)"""
        L.require_device(ids_flat, 'indices')
        M = ids_flat.numel() + (ids_prime.numel() if ids_prime is not None else 0)
        out = torch.empty((M, self.dim), device=ids_flat.device, dtype=torch.float32)
        ids = ids_flat if ids_flat.is_contiguous() else ids_flat.contiguous()
        L.lfq_decode(ids, self.project_out.weight, self.project_out.bias, out, M, self.dim, self.codebook_dim, ids_prime=ids_prime, perm=perm)
        return out

    def indices_to_codes(self, indices, project_out=True):
        assert project_out, 'raw +-1 codes are not exposed by the MI355X build'
        is_img_or_video = indices.ndim >= 3
        codes = self.codes_2d(indices.reshape(-1).long()).reshape(*indices.shape, self.dim)
        if is_img_or_video:
            codes = codes.movedim(-1, 1)
        return codes

    def aux_config(self, inv_temperature=100.):
        """keyword arguments of _lib.lfq_aux for this module (published forward default inv_temperature = 100)"""
        if not 2 <= self.codebook_dim <= 16:
            # the pk_lfq_aux_* kernels factorise the 2^cd codebook distribution over two halves of <= 8 bits each (256-entry LDS tables)
            raise ValueError(f'the LFQ training-mode auxiliary loss (entropy + commitment, pk_lfq_aux_*) is built for codebook sizes 4 .. 65536; '
                             f'codebook_size = {self.codebook_size} (codebook_dim {self.codebook_dim}) can be run for inference and for the '
                             'reconstruction-only objective (use_vgg_and_gan=False) only')
        return dict(inv_temperature=inv_temperature, codebook_scale=self.codebook_scale, entropy_loss_weight=self.entropy_loss_weight,
                    commitment_loss_weight=self.commitment_loss_weight, diversity_gamma=self.diversity_gamma)

    def forward(self, x, inv_temperature=100., **_unused):
        """(b, n, dim) -> (quantized (b, n, dim), indices (b, n) int64, aux_loss).  The published module's predicate is `self.training` alone
        (the same predicate as the tokenizer's training step, train_cvivit.py): eval mode -> the hard codes, NO straight-through
        gradient into project_in, aux = 0; training mode -> the straight-through output and the entropy + commitment auxiliary loss
        (differentiable under grad mode through train_cvivit._LFQFn / pk_lfq_aux_*; under no_grad the same VALUES without a graph)."""
        b, n, d = x.shape
        x2 = x.reshape(b * n, d).float().contiguous()
        if self.training:
            from .train_cvivit import _LFQFn
            q, aux, _ = _LFQFn.apply(x2, self.project_in.weight, self.project_in.bias, self.project_out.weight, self.project_out.bias,
                                     self.aux_config(inv_temperature))
            with torch.no_grad():
                ids = self.encode_ids(x2.detach())
            return q.reshape(b, n, d), ids.reshape(b, n), aux
        ids = self.encode_ids(x2)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.project_out.parameters()):
            from .train_cvivit import _LFQFn
            q = _LFQFn.apply(x2, self.project_in.weight, self.project_in.bias, self.project_out.weight, self.project_out.bias, None, False)
        else:
            q = self.codes_2d(ids)
        return q.reshape(b, n, d), ids.reshape(b, n), torch.zeros((), device=x.device)


class _CosineSimCodebook(nn.Module):
    """holder mirroring the library's `_codebook` sub-module: `embed` is (1, codebook_size, dim), unit-norm rows."""

    def __init__(self, dim, codebook_size):
        super().__init__()
        embed = torch.nn.functional.normalize(torch.randn(1, codebook_size, dim), dim=-1)
        self.register_buffer('initted', torch.tensor([True]))
        self.register_buffer('cluster_size', torch.zeros(1, codebook_size))
        self.register_buffer('embed_avg', embed.clone())
        self.register_buffer('embed', embed)


class VectorQuantize(nn.Module):
    """`vector_quantize_pytorch.VectorQuantize(dim, codebook_size, use_cosine_sim=True)` at inference
    (reference call sites cvivit.py:321 construct, :568-570 forward with `mask=`, :441 `.codebook[indices]`; the
    package is not vendored -- published eval behaviour: l2-normalise the input, argmax of its dot product with the
    unit-norm codebook, gather; restated in oracle/lfq.py).  The lookup is the vocab-head kernel in its no-noise mode
    (one fused GEMM + argmax, the (M, 65536) similarity matrix is never written) and always runs in exact f32: ids are
    an argmax over 65 536 cosine similarities."""

    def __init__(self, *, dim, codebook_size, use_cosine_sim=True, **_unused):
        super().__init__()
        assert use_cosine_sim, 'only the cosine-sim codebook (the reference construction, cvivit.py:321) is built'
        assert dim % 32 == 0 and codebook_size % 4 == 0
        self.dim, self.codebook_size = dim, codebook_size
        self._codebook = _CosineSimCodebook(dim, codebook_size)

    @property
    def codebook(self):
        return self._codebook.embed[0]

    def encode_ids(self, x2d, return_proj=False):
        assert not return_proj, 'the margin audit projection exists for LFQ only'
        L.require_device(x2d, 'tokens')
        M, D = x2d.shape
        V = self.codebook_size
        xn = torch.empty_like(x2d)
        L.l2norm_rows(x2d, xn, M, D)
        cb = self.codebook.contiguous()
        zero_bias = torch.zeros((V,), device=x2d.device, dtype=torch.float32)
        partials = torch.empty((5 * L.vocab_ntiles(V) * M,), device=x2d.device, dtype=torch.float32)
        L.vocab_sample(L.F32, xn, cb, zero_bias, M, V, D, 1.0, None, None, 0, False, partials, no_noise=True)
        ids = torch.empty((M,), device=x2d.device, dtype=torch.int64)
        L.vocab_reduce(partials, M, V, None, None, None, ids, None, False)
        return ids

    def codes_2d(self, ids_flat, *, ids_prime=None, perm=(0, 0)):
        if ids_prime is not None:
            ids_flat = torch.cat((ids_prime, ids_flat), dim=-1)
        codes = self.codebook.index_select(0, ids_flat.reshape(-1).long())
        if perm[0]:
            pb, pc = perm
            codes = codes.view(-1, pb, pc, codes.shape[-1]).transpose(1, 2).reshape(-1, codes.shape[-1])
        return codes

    def forward(self, x, mask=None, **_unused):
        b, n, d = x.shape
        x2 = x.reshape(b * n, d).float().contiguous()
        ids = self.encode_ids(x2)
        q = self.codes_2d(ids)
        return q.reshape(b, n, d), ids.reshape(b, n), torch.zeros((), device=x.device)
