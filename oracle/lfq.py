"""CPU restatement of the quantizers the reference imports from the un-vendored
``vector-quantize-pytorch`` package (reference setup.py:33, pin ``>=1.11.8``).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Call sites in the reference:
  phenaki_pytorch/cvivit.py:17   import VectorQuantize, LFQ
  phenaki_pytorch/cvivit.py:319  LFQ(dim = dim, codebook_size = codebook_size, **kwargs)
  phenaki_pytorch/cvivit.py:321  VectorQuantize(dim, codebook_size, use_cosine_sim = True)
  phenaki_pytorch/cvivit.py:570  tokens, indices, aux = self.vq(tokens)
  phenaki_pytorch/cvivit.py:439  self.vq.indices_to_codes(indices)
  phenaki_pytorch/cvivit.py:441  self.vq.codebook[indices]

Published behaviour restated (eval mode, num_codebooks = 1), SURVEY.md 8c:
  codebook_dim = log2(codebook_size); project_in = Linear(dim, codebook_dim),
  project_out = Linear(codebook_dim, dim) (both with bias) when dim != codebook_dim;
  buffer mask = 2 ** arange(codebook_dim - 1, -1, -1)  (MSB first);
  forward: x = project_in(x); q = where(x > 0, +1, -1); indices = sum((q > 0) * mask) (int64);
  returns (project_out(q), indices, aux_loss = 0);
  training mode (module.train()): the straight-through estimator, q <- x + (q - x).detach(), so d project_in(x) = d q.  The entropy /
  commitment auxiliary loss of the published module is NOT restated (returned as 0): the reference consumes it only in the GAN branch
  (cvivit.py:667), which is out of scope -- the use_vgg_and_gan = False step returns the reconstruction loss alone (cvivit.py:624-627);
  indices_to_codes(ids): bits = (ids[..., None] & mask) != 0; codes = bits * 2 - 1; project_out(codes).
PARITY UNPINNED: the upstream package is absent, no golden vectors exist for it.
"""
import math
import torch
import torch.nn.functional as F
from torch import nn


class LFQ(nn.Module):
    def __init__(self, *, dim, codebook_size, codebook_scale=1.0, **_ignored):
        super().__init__()
        cd = int(math.log2(codebook_size))
        assert 2 ** cd == codebook_size, 'codebook size must be a power of two'
        self.dim, self.codebook_dim, self.codebook_size = dim, cd, codebook_size
        self.codebook_scale = codebook_scale
        has_proj = dim != cd
        self.project_in = nn.Linear(dim, cd) if has_proj else nn.Identity()
        self.project_out = nn.Linear(cd, dim) if has_proj else nn.Identity()
        self.register_buffer('mask', 2 ** torch.arange(cd - 1, -1, -1))

    def indices_to_codes(self, indices, project_out=True):
        is_img_or_video = indices.ndim >= 3
        bits = ((indices[..., None].long() & self.mask) != 0).float()
        codes = bits * self.codebook_scale * 2 - self.codebook_scale
        if project_out:
            codes = self.project_out(codes)
        if is_img_or_video:
            codes = codes.movedim(-1, 1)
        return codes

    def forward(self, x, **_ignored):
        x = self.project_in(x)
        scale = torch.full_like(x, self.codebook_scale)
        q = torch.where(x > 0, scale, -scale)
        indices = ((q > 0).long() * self.mask.long()).sum(dim=-1)
        if self.training:
            q = x + (q - x).detach()
        return self.project_out(q), indices, torch.zeros((), device=x.device)


class VectorQuantize(nn.Module):
    """cosine-sim codebook lookup (eval only): l2-normalise input and codebook,
    argmax of their dot product, gather.  ``.codebook`` is (codebook_size, dim)."""

    def __init__(self, *, dim, codebook_size, use_cosine_sim=True, **_ignored):
        super().__init__()
        assert use_cosine_sim
        self.embed = nn.Parameter(F.normalize(torch.randn(codebook_size, dim), dim=-1))

    @property
    def codebook(self):
        return self.embed

    def forward(self, x, mask=None):
        xn = F.normalize(x, dim=-1)
        cb = F.normalize(self.embed, dim=-1)
        indices = (xn @ cb.t()).argmax(dim=-1)
        return cb[indices], indices, torch.zeros((), device=x.device)
