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
    def __init__(self, *, dim, codebook_size, codebook_scale=1.0, **_unused):
        super().__init__()
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

    def codes_2d(self, ids_flat):
        """ids (M,) int64 -> project_out(+-1 codes) (M, dim) f32"""
        L.require_device(ids_flat, 'indices')
        M = ids_flat.numel()
        out = torch.empty((M, self.dim), device=ids_flat.device, dtype=torch.float32)
        L.lfq_decode(ids_flat.contiguous(), self.project_out.weight, self.project_out.bias, out, M, self.dim, self.codebook_dim)
        return out

    def indices_to_codes(self, indices, project_out=True):
        assert project_out, 'raw +-1 codes are not exposed by the MI355X build'
        is_img_or_video = indices.ndim >= 3
        codes = self.codes_2d(indices.reshape(-1).long()).reshape(*indices.shape, self.dim)
        if is_img_or_video:
            codes = codes.movedim(-1, 1)
        return codes

    def forward(self, x, **_unused):
        """(b, n, dim) -> (quantized (b, n, dim), indices (b, n) int64, aux_loss 0)  [eval semantics]"""
        b, n, d = x.shape
        x2 = x.reshape(b * n, d).float().contiguous()
        ids = self.encode_ids(x2)
        q = self.codes_2d(ids)
        return q.reshape(b, n, d), ids.reshape(b, n), torch.zeros((), device=x.device)
