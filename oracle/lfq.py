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
  training mode (module.train()): the straight-through estimator, q <- x + (q - x).detach(), so d project_in(x) = d q, AND the auxiliary
  loss (third return; consumed by the GAN generator objective, cvivit.py:570 + :666 `loss = recon_loss + perceptual_loss + vq_aux_loss + ...`),
  restated from the published `LFQ.forward` (lookup_free_quantization.py of the pinned line, constructor defaults entropy_loss_weight = 0.1,
  commitment_loss_weight = 0.25, diversity_gamma = 1., forward default inv_temperature = 100.; kwargs arrive through
  `lookup_free_quantization_kwargs`, cvivit.py:319):
      codebook  = all 2^cd sign vectors * codebook_scale (MSB-first bit order of `mask`)
      distance  = -2 * einsum('... i d, j d -> ... i j', original_input, codebook)            # "the same as euclidean distance up to a constant"
      prob      = (-distance * inv_temperature).softmax(dim = -1)
      per_sample_entropy = entropy(prob).mean();   entropy(p) = (-p * log(p.clamp(min = 1e-5))).sum(-1)
      avg_prob  = mean of prob over every token of the batch;  codebook_entropy = entropy(avg_prob).mean()
      entropy_aux_loss = per_sample_entropy - diversity_gamma * codebook_entropy
      commit_loss = F.mse_loss(original_input, quantized.detach())
      aux_loss  = entropy_aux_loss * entropy_loss_weight + commit_loss * commitment_loss_weight
  (the reference passes no `mask` to the LFQ, cvivit.py:568: every token counts).  `lfq_aux_loss` below materialises prob in token chunks, as the
  published code does in one piece;
  indices_to_codes(ids): bits = (ids[..., None] & mask) != 0; codes = bits * 2 - 1; project_out(codes).
PARITY UNPINNED: the upstream package is absent, no golden vectors exist for it.
"""
import math
import torch
import torch.nn.functional as F
from torch import nn


LFQ_DEFAULTS = dict(entropy_loss_weight=0.1, commitment_loss_weight=0.25, diversity_gamma=1., inv_temperature=100.)
LOG_EPS = 1e-5


def lfq_entropy(prob):
    """published `entropy`: (-prob * log(prob.clamp(min = 1e-5))).sum(-1)"""
    return (-prob * prob.clamp(min=LOG_EPS).log()).sum(dim=-1)


def lfq_aux_loss(original_input, *, codebook_scale=1.0, entropy_loss_weight=0.1, commitment_loss_weight=0.25, diversity_gamma=1.,
                 inv_temperature=100., chunk=512, breakdown=None):
    """the published training-mode auxiliary loss of LFQ.forward for `original_input` = project_in(x), (..., cd); differentiable (autograd).
    breakdown (dict): receives per_sample_entropy / codebook_entropy / commitment."""
    cd = original_input.shape[-1]
    z = original_input.reshape(-1, cd)
    n = z.shape[0]
    mask = 2 ** torch.arange(cd - 1, -1, -1)
    bits = ((torch.arange(2 ** cd)[:, None] & mask) != 0).to(z.dtype)
    codebook = bits * codebook_scale * 2 - codebook_scale                                   # (2^cd, cd), row j = the code of index j
    per_sample, avg = z.new_zeros(()), z.new_zeros(2 ** cd)
    for i0 in range(0, n, chunk):
        distance = -2 * torch.einsum('i d, j d -> i j', z[i0:i0 + chunk], codebook)
        prob = (-distance * inv_temperature).softmax(dim=-1)
        per_sample = per_sample + lfq_entropy(prob).sum()
        avg = avg + prob.sum(dim=0)
    per_sample_entropy = per_sample / n
    codebook_entropy = lfq_entropy(avg / n)
    quantized = torch.where(z > 0, torch.full_like(z, codebook_scale), torch.full_like(z, -codebook_scale))
    commit = F.mse_loss(z, quantized.detach())
    if breakdown is not None:
        breakdown.update(per_sample_entropy=per_sample_entropy.detach(), codebook_entropy=codebook_entropy.detach(), commitment=commit.detach())
    return (per_sample_entropy - diversity_gamma * codebook_entropy) * entropy_loss_weight + commit * commitment_loss_weight


class LFQ(nn.Module):
    def __init__(self, *, dim, codebook_size, codebook_scale=1.0, entropy_loss_weight=0.1, commitment_loss_weight=0.25, diversity_gamma=1.,
                 **_ignored):
        super().__init__()
        self.entropy_loss_weight, self.commitment_loss_weight, self.diversity_gamma = entropy_loss_weight, commitment_loss_weight, diversity_gamma
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

    def forward(self, x, inv_temperature=100., **_ignored):
        x = self.project_in(x)
        scale = torch.full_like(x, self.codebook_scale)
        q = torch.where(x > 0, scale, -scale)
        indices = ((q > 0).long() * self.mask.long()).sum(dim=-1)
        aux = torch.zeros((), device=x.device)
        if self.training:
            aux = lfq_aux_loss(x, codebook_scale=self.codebook_scale, entropy_loss_weight=self.entropy_loss_weight,
                               commitment_loss_weight=self.commitment_loss_weight, diversity_gamma=self.diversity_gamma, inv_temperature=inv_temperature)
            q = x + (q - x).detach()
        return self.project_out(q), indices, aux


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
