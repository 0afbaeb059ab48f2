"""Text-encoder boundary (reference /root/reference/phenaki_pytorch/t5.py:64-103) and the T5 v1.1 encoder on the MI355X kernels.

`Phenaki.sample` consumes a (B, L, d) f32 context whose padded positions are zero-filled (the text mask is `any(embeds != 0)`,
phenaki_pytorch.py:461).  `t5_encode_text` keeps the reference's contract: tokenizer + checkpoint come from HuggingFace transformers
(neither can be downloaded offline), and the `Phenaki.encode_texts` instance attribute stays the injection point for cached /
precomputed embeddings (`phenaki.encode_texts = lambda texts, output_device=None: cached`).

SURVEY.md 8f row 2 -- `T5Encoder`: the encoder stack itself (what `T5EncoderModel(...).last_hidden_state` computes in t5.py:86-89) with
HuggingFace's module tree / state_dict keys, its forward on libphenaki_hip.so: pk_rmsnorm (T5LayerNorm), pk_gemm (q|k|v in one GEMM, o,
[wi_0 ; wi_1], wo + residual), pk_attn_prep in its plain mode (no l2norm, no scaling) + pk_attn_fwd with the bucketed relative-position
bias and the key mask, pk_gated_gelu_tanh (gelu_new gate).  `register_t5_encoder(name, encoder)` makes `t5_encode_text` run the HIP
encoder behind HF's tokenizer.  dim_head (d_kv) must be 64 (every T5 v1.1 size).  Checked against the real HF module on name-keyed
random weights (tests/golden/t5_*.pt).
"""
import math

import torch
from torch import nn

from . import _lib as L
from .attention import PackedModule, _cache, compute_dtype_of, linear_weight, pack_linear_weight

DEFAULT_T5_NAME = 'google/t5-v1_1-base'
MAX_LENGTH = 256

_KNOWN_DIMS = {
    't5-small': 512, 't5-base': 768, 't5-large': 1024, 't5-3b': 1024, 't5-11b': 1024,
    'google/t5-v1_1-small': 512, 'google/t5-v1_1-base': 768, 'google/t5-v1_1-large': 1024,
    'google/t5-v1_1-xl': 2048, 'google/t5-v1_1-xxl': 4096,
}
_MODELS = {}


def get_encoded_dim(name):
    if name in _KNOWN_DIMS:
        return _KNOWN_DIMS[name]
    from transformers import T5Config
    return T5Config.from_pretrained(name).d_model


def _get(name):
    if name not in _MODELS:
        from transformers import T5EncoderModel, T5Tokenizer
        _MODELS[name] = (T5EncoderModel.from_pretrained(name).eval(), T5Tokenizer.from_pretrained(name))
    return _MODELS[name]


_HIP_ENCODERS = {}


def register_t5_encoder(name, encoder):
    """run `encoder` (a T5Encoder on a HIP device, holding the checkpoint's weights) for `t5_encode_text(texts, name=name)`; the
    tokenizer stays HuggingFace's.  None removes the registration."""
    if encoder is None:
        _HIP_ENCODERS.pop(name, None)
    else:
        _HIP_ENCODERS[name] = encoder


@torch.no_grad()
def t5_encode_text(texts, name=DEFAULT_T5_NAME, output_device=None):
    """List[str] -> (B, L, d) f32, pad positions zero-filled (t5.py:97-100)."""
    hip = _HIP_ENCODERS.get(name)
    if hip is not None:
        from transformers import T5Tokenizer
        key = ('tok', name)
        if key not in _MODELS:                       # load the sentencepiece model once, not on every call
            _MODELS[key] = T5Tokenizer.from_pretrained(name)
        tok = _MODELS[key]
        enc = tok.batch_encode_plus(texts, return_tensors='pt', padding='longest', max_length=MAX_LENGTH, truncation=True)
        device = next(hip.parameters()).device
        out = hip(enc.input_ids.to(device), enc.attention_mask.to(device))
        return out if output_device is None else out.to(output_device)
    model, tok = _get(name)
    if torch.cuda.is_available():
        model = model.cuda()
    device = next(model.parameters()).device
    enc = tok.batch_encode_plus(texts, return_tensors='pt', padding='longest', max_length=MAX_LENGTH, truncation=True)
    input_ids, attn_mask = enc.input_ids.to(device), enc.attention_mask.to(device)
    out = model(input_ids=input_ids, attention_mask=attn_mask).last_hidden_state.detach()
    out = out.masked_fill(~attn_mask.bool()[..., None], 0.)
    return out if output_device is None else out.to(output_device)


# ------------------------------------------------------------------------------------------ the encoder on the MI355X kernels
# module tree = HuggingFace's T5EncoderModel (transformers models/t5/modeling_t5.py), so `load_state_dict(hf_model.state_dict())` works

class _T5LayerNorm(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.variance_epsilon = eps


class _T5Attention(nn.Module):
    def __init__(self, d, inner, heads, buckets, has_bias):
        super().__init__()
        self.q, self.k, self.v = (nn.Linear(d, inner, bias=False) for _ in range(3))
        self.o = nn.Linear(inner, d, bias=False)
        if has_bias:
            self.relative_attention_bias = nn.Embedding(buckets, heads)


class _T5LayerSelfAttention(nn.Module):
    def __init__(self, d, inner, heads, buckets, has_bias, eps):
        super().__init__()
        self.SelfAttention = _T5Attention(d, inner, heads, buckets, has_bias)
        self.layer_norm = _T5LayerNorm(d, eps)


class _T5DenseGatedActDense(nn.Module):
    def __init__(self, d, d_ff):
        super().__init__()
        self.wi_0, self.wi_1 = nn.Linear(d, d_ff, bias=False), nn.Linear(d, d_ff, bias=False)
        self.wo = nn.Linear(d_ff, d, bias=False)


class _T5LayerFF(nn.Module):
    def __init__(self, d, d_ff, eps):
        super().__init__()
        self.DenseReluDense = _T5DenseGatedActDense(d, d_ff)
        self.layer_norm = _T5LayerNorm(d, eps)


class _T5Block(nn.Module):
    def __init__(self, d, inner, heads, d_ff, buckets, has_bias, eps):
        super().__init__()
        self.layer = nn.ModuleList([_T5LayerSelfAttention(d, inner, heads, buckets, has_bias, eps), _T5LayerFF(d, d_ff, eps)])


class _T5Stack(nn.Module):
    def __init__(self, embed, d, inner, heads, d_ff, layers, buckets, eps):
        super().__init__()
        self.embed_tokens = embed
        self.block = nn.ModuleList([_T5Block(d, inner, heads, d_ff, buckets, i == 0, eps) for i in range(layers)])
        self.final_layer_norm = _T5LayerNorm(d, eps)


def _relative_position_bucket(rel, num_buckets, max_distance):
    """modeling_t5.py T5Attention._relative_position_bucket, bidirectional"""
    num_buckets //= 2
    buckets = (rel > 0).long() * num_buckets
    rp = rel.abs()
    max_exact = num_buckets // 2
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(rp < max_exact, rp, large)


class T5Encoder(PackedModule):
    """T5 v1.1 encoder (gated-GELU feed-forward) with HuggingFace T5EncoderModel's constructor config names and state_dict keys."""

    def __init__(self, *, vocab_size=32128, d_model=768, d_kv=64, d_ff=2048, num_layers=12, num_heads=12,
                 relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                 feed_forward_proj='gated-gelu', **_unused):
        super().__init__()
        assert d_kv == 64, 'the MI355X attention kernels are built for a head width of 64 (every T5 v1.1 size has d_kv = 64)'
        assert feed_forward_proj == 'gated-gelu', 'T5 v1.1 (gated-GELU feed-forward) is what the reference loads (t5.py:18)'
        self.d_model, self.heads, self.d_ff = d_model, num_heads, d_ff
        self.num_buckets, self.max_distance = relative_attention_num_buckets, relative_attention_max_distance
        self.shared = nn.Embedding(vocab_size, d_model)
        self.encoder = _T5Stack(self.shared, d_model, num_heads * d_kv, num_heads, d_ff, num_layers, relative_attention_num_buckets,
                                layer_norm_epsilon)

    @classmethod
    def from_hf(cls, hf_model):
        """a T5Encoder holding the weights of a HuggingFace T5EncoderModel"""
        m = cls(**hf_model.config.to_dict())
        m.load_state_dict(hf_model.state_dict())
        return m

    def _position_bias(self, Ln, device):
        table = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight

        def build():
            ctx = torch.arange(Ln, device=device)[:, None]
            mem = torch.arange(Ln, device=device)[None, :]
            b = _relative_position_bucket(mem - ctx, self.num_buckets, self.max_distance)
            return table.detach().float()[b].permute(2, 0, 1).contiguous()                  # (heads, L, L)
        return _cache(self).get(('pos_bias', Ln), [table], build)

    def _cat_weight(self, owner, key, lins, dt):
        return _cache(owner).get((key, dt), [l.weight for l in lins],
                                 lambda: pack_linear_weight(torch.cat([l.weight.detach().float() for l in lins], dim=0), dt))

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, zero_pads=True):
        """(B, L) int64 ids [, (B, L) mask] -> last_hidden_state (B, L, d_model) f32; zero_pads: masked positions zero-filled (t5.py:97-100)"""
        L.require_device(input_ids, 'input_ids')
        dt = compute_dtype_of(self)
        td = L.tdtype(dt)
        B, Ln = input_ids.shape
        M, d, h = B * Ln, self.d_model, self.heads
        inner = h * 64
        dev = input_ids.device
        km = None if attention_mask is None else attention_mask.to(torch.uint8).contiguous()
        x = self.shared.weight.detach().float().index_select(0, input_ids.reshape(-1))         # (M, d) f32 (the text encoder runs once per prompt)
        bias = self._position_bias(Ln, dev)
        nq_pad, nk_pad = L.attn_pads(Ln, Ln, 0)
        for blk in self.encoder.block:
            sa, ff = blk.layer[0], blk.layer[1]
            att, dense = sa.SelfAttention, ff.DenseReluDense
            xn = torch.empty((M, d), device=dev, dtype=td)
            L.rmsnorm(x, sa.layer_norm.weight, M, d, xn, eps=sa.layer_norm.variance_epsilon)
            qkv = torch.empty((M, 3 * inner), device=dev, dtype=torch.float32)
            L.gemm(dt, xn, self._cat_weight(att, 'qkv', (att.q, att.k, att.v), dt), M, 3 * inner, d, C=qkv)
            Qp = torch.empty((B * h * nq_pad * 64,), device=dev, dtype=td)
            Kp = torch.empty((B * h * nk_pad * 64,), device=dev, dtype=td)
            Vt = torch.empty((B * h * nk_pad * 64,), device=dev, dtype=td)
            L.attn_prep(dt, qkv[:, :inner], qkv[:, inner:], None, None, None, 1.0, Qp, Kp, Vt, B, h, Ln, Ln, 0)      # plain mode: no l2norm, no scale
            o = torch.empty((M, inner), device=dev, dtype=td)
            L.attn_fwd(dt, Qp, Kp, Vt, o, B, h, Ln, Ln, 0, bias=bias, kmask=km)
            x2 = torch.empty_like(x)
            L.gemm(dt, o, linear_weight(att.o, dt), M, d, inner, C=x2, res=x)
            x = x2
            L.rmsnorm(x, ff.layer_norm.weight, M, d, xn, eps=ff.layer_norm.variance_epsilon)
            hbuf = torch.empty((M, 2 * self.d_ff), device=dev, dtype=torch.float32)
            L.gemm(dt, xn, self._cat_weight(dense, 'wi', (dense.wi_0, dense.wi_1), dt), M, 2 * self.d_ff, d, C=hbuf)
            hm = torch.empty((M, self.d_ff), device=dev, dtype=td)
            L.gated_gelu_tanh(hbuf, hm, M, self.d_ff)
            x2 = torch.empty_like(x)
            L.gemm(dt, hm, linear_weight(dense.wo, dt), M, d, self.d_ff, C=x2, res=x)
            x = x2
        out = torch.empty((M, d), device=dev, dtype=torch.float32)
        fl = self.encoder.final_layer_norm
        L.rmsnorm(x, fl.weight, M, d, out, eps=fl.variance_epsilon, rowmask=km.reshape(-1) if (zero_pads and km is not None) else None)
        return out.view(B, Ln, d)
