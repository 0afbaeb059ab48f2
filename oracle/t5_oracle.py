"""CPU restatement of the T5 v1.1 ENCODER the reference calls for its text context (/root/reference/phenaki_pytorch/t5.py:64-103:
`T5EncoderModel.from_pretrained(name)(input_ids, attention_mask).last_hidden_state`, pads zero-filled at :97-100).

TEST INFRASTRUCTURE ONLY.  The algorithm lives in a third-party dependency that is not vendored in /root/reference: HuggingFace
`transformers` (the reference's setup.py lists it unpinned; the build image carries transformers 5.15.0), models/t5/modeling_t5.py.
It is restated here from that published implementation -- T5LayerNorm (RMS norm, no bias), T5Attention (no 1/sqrt(d) scaling, a
bucketed relative position bias owned by block 0 and shared by all blocks), T5DenseGatedActDense (gelu_new(wi_0 x) * wi_1 x) -- and
PINNED by tests/test_oracle_golden.py against outputs of the real HF module on name-keyed random weights (oracle/make_golden.py
t5_golden; no checkpoint can be downloaded offline).
"""
import math

import torch

T5_TINY = dict(vocab_size=512, d_model=128, d_kv=64, num_heads=2, d_ff=256, num_layers=2, relative_attention_num_buckets=32,
               relative_attention_max_distance=128, feed_forward_proj='gated-gelu', layer_norm_epsilon=1e-6)
# google/t5-v1_1-base (the reference's DEFAULT_T5_NAME, t5.py:18) layer geometry.  The golden runs 4 of its 12 layers: with RANDOM weights
# (no checkpoint offline) the unscaled dot-product attention amplifies f32 round-off by ~1.7x per layer -- HF's own module and this
# restatement, both f32, already differ by 2e-3 after 11 layers (2.5e-6 after one) -- so a 12-layer random-weight golden pins nothing at
# 1e-3; the per-layer arithmetic is identical at any depth.
T5_BASE = dict(vocab_size=32128, d_model=768, d_kv=64, num_heads=12, d_ff=2048, num_layers=4, relative_attention_num_buckets=32,
               relative_attention_max_distance=128, feed_forward_proj='gated-gelu', layer_norm_epsilon=1e-6)


def relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """modeling_t5.py T5Attention._relative_position_bucket, bidirectional = True (encoder)"""
    num_buckets //= 2
    buckets = (relative_position > 0).long() * num_buckets
    rp = relative_position.abs()
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rp, large)


def position_bias(table, L, num_buckets=32, max_distance=128):
    """(heads, L, L): table (num_buckets, heads) = block 0's relative_attention_bias.weight; bias[h][i][j] for query i, key j"""
    ctx = torch.arange(L)[:, None]
    mem = torch.arange(L)[None, :]
    b = relative_position_bucket(mem - ctx, num_buckets, max_distance)
    return table[b].permute(2, 0, 1).contiguous()


def rmsnorm(x, w, eps=1e-6):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def t5_encode(sd, cfg, input_ids, attention_mask, zero_pads=True):
    """last_hidden_state (B, L, d) of the encoder on the HF state_dict `sd`; zero_pads: the masked_fill of t5.py:97-100"""
    h_, dk = cfg['num_heads'], cfg['d_kv']
    B, L = input_ids.shape
    x = sd['encoder.embed_tokens.weight'][input_ids]
    bias = position_bias(sd['encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight'], L,
                         cfg['relative_attention_num_buckets'], cfg['relative_attention_max_distance'])
    neg = torch.finfo(torch.float32).min
    ext = (1.0 - attention_mask.float())[:, None, None, :] * neg
    for i in range(cfg['num_layers']):
        p = f'encoder.block.{i}.layer.'
        n = rmsnorm(x, sd[p + '0.layer_norm.weight'], cfg['layer_norm_epsilon'])
        q = (n @ sd[p + '0.SelfAttention.q.weight'].t()).view(B, L, h_, dk).transpose(1, 2)
        k = (n @ sd[p + '0.SelfAttention.k.weight'].t()).view(B, L, h_, dk).transpose(1, 2)
        v = (n @ sd[p + '0.SelfAttention.v.weight'].t()).view(B, L, h_, dk).transpose(1, 2)
        scores = q @ k.transpose(-1, -2) + bias[None] + ext                      # no 1/sqrt(d) in T5
        o = (scores.softmax(-1) @ v).transpose(1, 2).reshape(B, L, h_ * dk)
        x = x + o @ sd[p + '0.SelfAttention.o.weight'].t()
        n = rmsnorm(x, sd[p + '1.layer_norm.weight'], cfg['layer_norm_epsilon'])
        ff = gelu_new(n @ sd[p + '1.DenseReluDense.wi_0.weight'].t()) * (n @ sd[p + '1.DenseReluDense.wi_1.weight'].t())
        x = x + ff @ sd[p + '1.DenseReluDense.wo.weight'].t()
    x = rmsnorm(x, sd['encoder.final_layer_norm.weight'], cfg['layer_norm_epsilon'])
    if zero_pads:
        x = x.masked_fill(~attention_mask.bool()[..., None], 0.)
    return x


def t5_state_dict(cfg, salt=5):
    """name-keyed random weights with the HF T5EncoderModel key set of `cfg` (no module needed; the tied embedding keeps the value of
    the LAST key written, `encoder.embed_tokens.weight`, exactly as weights.fill_module does on the tied tensor)"""
    from oracle import weights
    d, inner, F, V = cfg['d_model'], cfg['num_heads'] * cfg['d_kv'], cfg['d_ff'], cfg['vocab_size']
    shapes = {'shared.weight': (V, d), 'encoder.embed_tokens.weight': (V, d)}
    for i in range(cfg['num_layers']):
        p = f'encoder.block.{i}.layer.'
        for nm in ('q', 'k', 'v'):
            shapes[p + f'0.SelfAttention.{nm}.weight'] = (inner, d)
        shapes[p + '0.SelfAttention.o.weight'] = (d, inner)
        if i == 0:
            shapes[p + '0.SelfAttention.relative_attention_bias.weight'] = (cfg['relative_attention_num_buckets'], cfg['num_heads'])
        shapes[p + '0.layer_norm.weight'] = (d,)
        shapes[p + '1.DenseReluDense.wi_0.weight'] = (F, d)
        shapes[p + '1.DenseReluDense.wi_1.weight'] = (F, d)
        shapes[p + '1.DenseReluDense.wo.weight'] = (d, F)
        shapes[p + '1.layer_norm.weight'] = (d,)
    shapes['encoder.final_layer_norm.weight'] = (d,)
    sd = {k: weights.fill_value(k, torch.empty(s), salt) for k, s in shapes.items()}
    sd['shared.weight'] = sd['encoder.embed_tokens.weight']
    return sd


def t5_inputs(cfg, B, L, seed=0):
    g = torch.Generator().manual_seed(4000 + seed)
    ids = torch.randint(2, cfg['vocab_size'], (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.bool)
    for b in range(1, B):                                                          # ragged lengths: pads (id 0) at the end
        n = max(1, L - 3 * b)
        mask[b, n:] = False
        ids[b, n:] = 0
    return ids, mask
