"""Transformer building blocks with the reference's module tree / state_dict keys
(/root/reference/phenaki_pytorch/attention.py:29-332) whose forward passes run on libphenaki_hip.so.

Every activation that lives in HBM between kernels is a 2-D f32 (rows, dim) buffer; the GEMM / attention
operands are bf16 (`compute_dtype = 'bf16'`, f32 accumulation) or exact f32 (`'fp32'`).
There is no eager fallback: tensors must be on a HIP device.
"""
import math
import os

import torch
from torch import nn

from . import _lib as L

_DTYPES = {'fp32': L.F32, 'float32': L.F32, 'f32': L.F32, 'bf16': L.BF16, 'bfloat16': L.BF16, 'bf16x3': L.BF16X3}


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


def resolve_dtype(name):
    if name not in _DTYPES:
        raise ValueError(f'compute dtype must be one of {sorted(_DTYPES)}, got {name!r}')
    return _DTYPES[name]


def set_compute_dtype(module, name):
    """'fp32' (exact f32 MFMA; the reference's default precision), 'bf16' (bf16 MFMA operands, f32 accumulation) or 'bf16x3'
    (split-bf16: f32 activations, every matrix product as three bf16 MFMAs on (hi, lo) operand planes -- ~1e-5 per product,
    held to the same tolerances as 'fp32' at a fraction of its cost)."""
    resolve_dtype(name)
    for m in module.modules():
        m._pk_compute_dtype = name
    return module


def compute_dtype_of(module):
    return resolve_dtype(getattr(module, '_pk_compute_dtype', 'fp32'))


def round_up(x, m):
    return (x + m - 1) // m * m


class PackCache:
    """device-side packed copies of a module's weights, rebuilt when a source parameter changes."""

    def __init__(self):
        self.store = {}

    def get(self, key, params, build):
        stamp = tuple((p.data_ptr(), p._version, p.device) for p in params)
        hit = self.store.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            val = build()
        self.store[key] = (stamp, val)
        return val


def _cache(module):
    c = module.__dict__.get('_pk_cache')
    if c is None:
        c = PackCache()
        module.__dict__['_pk_cache'] = c
    return c


def invalidate_packed(module):
    """drop every packed / cached device copy derived from `module`'s parameters (packed GEMM weights, PEG taps, position-bias
    tables, cached biases): they are rebuilt on the next call.  The caches key on (data_ptr, _version, device) of the source
    parameters, which catches optimizer steps, load_state_dict (also hooked below) and .to(); in-place writes through `.data`
    (`p.data.copy_`, `p.data.lerp_` -- ema-pytorch, weight surgery) do NOT bump `_version`: call this after them."""
    for m in module.modules():
        m.__dict__.pop('_pk_cache', None)
        # captured sampling hipGraphs (Phenaki.enable_sample_graph) hold raw pointers into the packed copies dropped above
        m.__dict__.pop('_pk_sample_graphs', None)
    return module


def param_fingerprint(module):
    """(data_ptr, _version) of every parameter and buffer: what a captured hipGraph of `module`'s kernels silently depends on"""
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


class PackedModule(nn.Module):
    """nn.Module whose packed-weight caches (and those of its children) are dropped by load_state_dict and by every _apply
    (.to / .cuda / .float ...)."""

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        invalidate_packed(self)
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        invalidate_packed(self)
        return out


class _NoBackward(torch.autograd.Function):
    """identity on a loss VALUE computed by the inference kernels (no autograd graph behind it): its backward raises instead of silently training nothing"""

    @staticmethod
    def forward(ctx, value, anchor, what):
        ctx.what = what
        return value.clone()

    @staticmethod
    def backward(ctx, grad):
        raise RuntimeError(f'{ctx.what} returned the VALUE of the objective through the inference kernels -- no backward kernels ran behind this '
                           'tensor.  The training steps are Phenaki.forward (train.py) and CViViT.forward (train_cvivit.py) with grad mode on and '
                           'trainable parameters of the module being trained')


def value_without_graph(module, what, value):
    """`value` as the caller of the reference signature expects it: under no_grad / frozen parameters the plain tensor; with grad
    mode on and trainable parameters a tensor that requires grad and whose .backward() raises a clear error (instead of either
    refusing the forward call, which broke `loss = phenaki(...)`, or silently returning a leaf that trains nothing)."""
    if not torch.is_grad_enabled():
        return value
    anchor = next((p for p in module.parameters() if p.requires_grad), None)
    if anchor is None:
        return value
    if isinstance(value, tuple):
        return (_NoBackward.apply(value[0], anchor, what),) + tuple(value[1:])
    return _NoBackward.apply(value, anchor, what)


def pack_linear_weight(w, dtype):
    """(N, K) f32 -> (N, Kpad) T with K zero-padded to the GEMM k-tile (64 bf16 / 32 f32): the LDS-DMA main loop
    relies on that zero padding to neutralise the K tail of A."""
    q = 64 if dtype == L.BF16 else 32
    n, k = w.shape
    kp = round_up(k, q)
    if dtype == L.F32 and kp == k and w.dtype == torch.float32 and w.is_contiguous():
        return w.detach()              # exact-f32 mode, K already a multiple of the k-tile: the LIVE weight, no copy to go stale
    out = torch.zeros((n, kp), device=w.device, dtype=L.tdtype(dtype))
    out[:, :k] = w.detach().to(out.dtype)
    if dtype == L.BF16X3:
        return L.split_planes(out)     # (hi | lo) bf16 planes per 32-element block, typed float32 (raw bits)
    return out


def linear_weight(lin, dtype):
    return _cache(lin).get(('w', dtype), [lin.weight], lambda: pack_linear_weight(lin.weight, dtype))


# LayerNorm folded into the nn.Linear that consumes it (bf16 mode; PK_LN_FOLD=0 keeps the separate LayerNorm launches for A/B):
#   LN(x) W^T = rstd * (x (gamma.W)^T - mean * s) + t,   s[n] = sum_k (gamma.W)[n][k],   t[n] = sum_k beta[k] W[n][k]
# The GEMM reads the bf16 copy of the residual stream that its producer wrote beside the f32 one and takes mean / rstd from its own
# A tiles (pk_gemm_ex ln_s / ln_t, pk_qkv_project / pk_qkv_attn q_ln_s): the LayerNorm launches and their 14 MB round trips go.
_LN_FOLD = os.environ.get('PK_LN_FOLD', '1') != '0'
# the feed-forward LayerNorm stays a separate launch by default: folded, the 128x128 FF1 kernel needs mean AND rstd of every row
# (64 v_dot2c per wave per k-tile beside its 16 MFMAs) and ran 25 -> 41 us at M = 4608 -- more than the LayerNorm launch it saves
# feed-forward LayerNorm: 0 = ln_rows launch in front of FF1; 1 = folded into FF1 with in-loop statistics (measured slower: FF1 25 -> 41 us);
# 2 = folded, statistics handed over by the to_out GEMM that wrote the rows (pk_gemm_ex stats_out -> ln_stats): no ln_rows launch.
# Measured (same-box A/B, 2 rounds): mode 2 vs 0  encode -5 %, decode -3 %, cfg sampling (M = 4608 rows) -3 %, but B = 8 sampling
# (M = 9216 rows: FF1 +3.6 us for the fold epilogue, ln_rows 6.3 us) +1 %  ->  the hand-over is used up to PK_LN_FOLD_FF_MAX_ROWS rows.
_LN_FOLD_FF = int(os.environ.get('PK_LN_FOLD_FF', '2'))
_LN_FOLD_FF_MAX_ROWS = int(os.environ.get('PK_LN_FOLD_FF_MAX_ROWS', '6144'))


def ff_fold_mode(rows):
    return _LN_FOLD_FF if (_LN_FOLD_FF_MAX_ROWS <= 0 or rows <= _LN_FOLD_FF_MAX_ROWS) else 0


# the split-bf16 (bf16x3) mode folds its LayerNorms the same way (round 4): the rows it reads ARE the f32 residual stream (no bf16 copy), the
# fused projection + attention kernels exist for it, and the same fold arithmetic runs on f32-grade products.  PK_LN_FOLD_X3=0: the round-3
# path (separate LayerNorm, q / kv GEMMs, pk_attn_prep, LDS-free short attention) for A/B timing.
_LN_FOLD_X3 = os.environ.get('PK_LN_FOLD_X3', '1') != '0'


def ln_fold_enabled(dtype):
    return _LN_FOLD and (dtype == L.BF16 or (dtype == L.BF16X3 and _LN_FOLD_X3))


def folded_weight(owner, key, w_f32, gamma, beta, dtype, params):
    """(packed gamma (.) W in T, s, t, beta_is_zero) for the (N, K) f32 weight `w_f32` (a callable building it), cached on `owner`"""
    def build():
        w = w_f32().float()
        wgf = w * gamma.detach().float()[None, :]
        wg = pack_linear_weight(wgf, dtype)
        if dtype == L.BF16:
            s = wg[:, :w.shape[1]].float().sum(dim=1).contiguous()        # from the ROUNDED operand the MFMAs see
        else:
            s = wgf.sum(dim=1).contiguous()                               # f32 / split-bf16 (hi + lo = the f32 value to 2^-17): the f32 row sums
        if beta is None:
            t, bz = torch.zeros_like(s), True
        else:
            t = (w @ beta.detach().float()).contiguous()
            bz = bool((beta.detach() == 0).all().item())
        return wg, s, t, bz
    return _cache(owner).get((key, dtype), params, build)


def f32c(t):
    return t.detach().float().contiguous()


# --------------------------------------------------------------------------- modules

class LayerNorm(nn.Module):
    """attention.py:29-36 : learned gamma, zero `beta` buffer (persistent -> in the state_dict)."""

    eps = 1e-5                      # F.layer_norm default, which attention.py:36 uses

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer('beta', torch.zeros(dim))

    def run(self, x2d, out=None, out2=None):
        M, D = x2d.shape
        L.layernorm(x2d, self.gamma, self.beta, M, D, out=out, out2=out2)

    def forward(self, x):
        L.require_device(x, 'x')
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        out = torch.empty_like(x2)
        self.run(x2, out=out)
        return out.reshape(x.shape)


class GEGLU(nn.Module):
    """attention.py:40-43 (placeholder module: the activation is fused into the first FeedForward GEMM's epilogue)."""

    def forward(self, x):
        raise RuntimeError('GEGLU runs fused inside FeedForward on the MI355X build')


class FeedForwardSeq(nn.Sequential):
    """attention.py:45-53 : Sequential(nn.LayerNorm, Linear(d, 2*inner, no bias), GEGLU, Dropout, Linear(inner, d, no bias)).
    run(): LN -> GEMM with interleaved (value, gate) rows + GEGLU epilogue -> GEMM + residual."""

    def _packed(self, dtype):
        lin1, lin2 = self[1], self[4]

        def build():
            inner = lin2.weight.shape[1]
            q = 8 if dtype == L.BF16 else 4
            ip = round_up(inner, q)
            w1 = lin1.weight.detach()
            d = w1.shape[1]
            w1p = torch.zeros((2 * ip, d), device=w1.device, dtype=torch.float32)
            w1p[0:2 * inner:2] = w1[:inner]          # value rows  (x, gate = chunk(2))
            w1p[1:2 * inner:2] = w1[inner:]          # gate rows
            w2p = torch.zeros((lin2.weight.shape[0], ip), device=w1.device, dtype=torch.float32)
            w2p[:, :inner] = lin2.weight.detach()
            return pack_linear_weight(w1p, dtype), pack_linear_weight(w2p, dtype), ip
        return _cache(self).get(('ff', dtype), [lin1.weight, lin2.weight], build)

    def _packed_folded(self, dtype):
        """first GEMM with the block's nn.LayerNorm folded in: (gamma (.) W1 interleaved / padded, s, t, beta_is_zero)"""
        lin1, lin2, ln = self[1], self[4], self[0]

        def w1():
            inner = lin2.weight.shape[1]
            ip = round_up(inner, 8 if dtype == L.BF16 else 4)
            w = lin1.weight.detach().float()
            w1p = torch.zeros((2 * ip, w.shape[1]), device=w.device, dtype=torch.float32)
            w1p[0:2 * inner:2] = w[:inner]
            w1p[1:2 * inner:2] = w[inner:]
            return w1p
        return folded_weight(self, 'ff1_ln', w1, ln.weight, ln.bias, dtype, [lin1.weight, ln.weight, ln.bias])

    def run(self, x2d, dtype, xt=None, want_t=False, stats=None):
        """x2d (M, D) f32 -> ff(x) + x  (M, D) f32.  xt: the T copy of x2d (bf16 mode, LayerNorm folded into the first GEMM);
        stats: the (M, D/32, 2) row statistics of xt its producer left (else the GEMM takes them from its own main loop);
        want_t: also return the T copy of the result for the next block -> (out, out_t)."""
        M, D = x2d.shape
        w1p, w2p, ip = self._packed(dtype)
        td = L.tdtype(dtype)
        ln = self[0]
        hmid = torch.empty((M, ip), device=x2d.device, dtype=td)
        if ln_fold_enabled(dtype) and ff_fold_mode(M):
            w1g, s1, t1, _ = self._packed_folded(dtype)
            if xt is None:
                xt = x2d.to(td)
            L.gemm(dtype, xt, w1g, M, 2 * ip, D, C=hmid, act=L.ACT_GEGLU, ln=(s1, t1, ln.eps), ln_stats=stats)
        else:
            xn = torch.empty((M, D), device=x2d.device, dtype=td)
            L.layernorm(x2d, ln.weight, ln.bias, M, D, out=xn, eps=ln.eps)
            L.gemm(dtype, xn, w1p, M, 2 * ip, D, C=hmid, act=L.ACT_GEGLU)
        out = torch.empty_like(x2d)
        out_t = torch.empty((M, D), device=x2d.device, dtype=td) if (want_t and dtype == L.BF16) else None
        L.gemm(dtype, hmid, w2p, M, D, ip, C=out, res=x2d, C2=out_t)
        return (out, out_t) if want_t else out

    def forward(self, x):
        L.require_device(x, 'x')
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        return (self.run(x2, compute_dtype_of(self)) - x2).reshape(x.shape)


def FeedForward(dim, mult=4, dropout=0.):
    inner_dim = int(mult * (2 / 3) * dim)
    return FeedForwardSeq(
        nn.LayerNorm(dim),
        nn.Linear(dim, inner_dim * 2, bias=False),
        GEGLU(),
        nn.Dropout(dropout),
        nn.Linear(inner_dim, dim, bias=False),
    )


class PEG(PackedModule):
    """attention.py:57-85 : depthwise Conv3d(dim, dim, 3, groups=dim) on the raw (b,t,h,w,d) reinterpretation."""

    def __init__(self, dim, causal=False):
        super().__init__()
        self.causal = causal
        self.dsconv = nn.Conv3d(dim, dim, 3, groups=dim)

    def _packed(self):
        w = self.dsconv.weight
        return _cache(self).get('wt', [w], lambda: w.detach().float().reshape(w.shape[0], 27).t().contiguous())

    def run(self, x2d, shape, want_t=False):
        """x2d (M, D) f32 contiguous, shape (b,t,h,w) with b*t*h*w == M -> peg(x) + x  [, its bf16 copy]"""
        b, t, h, w = shape
        M, D = x2d.shape
        assert b * t * h * w == M, 'PEG shape does not match the token buffer'
        out = torch.empty_like(x2d)
        out_t = torch.empty((M, D), device=x2d.device, dtype=torch.bfloat16) if want_t else None
        L.peg(x2d, self._packed(), self.dsconv.bias, out, b, t, h, w, D, self.causal, out_t=out_t)
        return (out, out_t) if want_t else out

    def forward(self, x, shape=None):
        L.require_device(x, 'x')
        needs_shape = x.ndim == 3
        assert not (needs_shape and not exists(shape))
        if not needs_shape:
            shape = tuple(x.shape[:-1])
        x2 = x.reshape(-1, x.shape[-1]).float().contiguous()
        return (self.run(x2, tuple(shape)) - x2).reshape(x.shape)


class AlibiPositionalBias(nn.Module):
    """attention.py:186-227 : per-head slopes; the bias itself is generated inside the attention kernel."""

    def __init__(self, heads):
        super().__init__()
        self.heads = heads
        slopes = torch.tensor(self._get_slopes(heads), dtype=torch.float32).reshape(heads, 1, 1)
        self.register_buffer('slopes', slopes, persistent=False)

    @staticmethod
    def _get_slopes(heads):
        def pow2_slopes(n):
            start = 2 ** (-2 ** -(math.log2(n) - 3))
            return [start * start ** i for i in range(n)]
        if math.log2(heads).is_integer():
            return pow2_slopes(heads)
        c = 2 ** math.floor(math.log2(heads))
        return pow2_slopes(c) + pow2_slopes(2 * c)[0::2][:heads - c]


class BiasSpec:
    """an attention bias that is a ContinuousPositionBias of a grid: the consumers pick the form they can use"""

    def __init__(self, module, dims):
        self.module, self.dims = module, dims

    @property
    def full(self):
        return self.module(*self.dims)

    def table(self):
        return self.module.table(*self.dims)


def _full_bias(b):
    return b.full if isinstance(b, BiasSpec) else b


class ContinuousPositionBias(PackedModule):
    """attention.py:229-275 : MLP over log-spaced relative positions -> (heads, n, n).
    The MLP always runs in exact f32 (the reference forces rel_pos.float()), and the result is cached per
    (weights, dims): the weights are frozen during sampling, the reference recomputes it every forward."""

    def __init__(self, *, dim, heads, num_dims=2, layers=2, log_dist=True, cache_rel_pos=False):
        super().__init__()
        assert log_dist, 'only log_dist = True (the reference default) is built'
        self.num_dims = num_dims
        self.log_dist = log_dist
        self.net = nn.ModuleList([])
        self.net.append(nn.Sequential(nn.Linear(num_dims, dim), nn.LeakyReLU(0.1)))
        for _ in range(layers - 1):
            self.net.append(nn.Sequential(nn.Linear(dim, dim), nn.LeakyReLU(0.1)))
        self.net.append(nn.Linear(dim, heads))
        self.cache_rel_pos = cache_rel_pos

    def forward(self, *dimensions, device=None):
        assert len(dimensions) == self.num_dims
        params = list(self.parameters())
        L.require_device(params[0], 'ContinuousPositionBias parameters')
        return _cache(self).get(('bias', tuple(dimensions)), params, lambda: self._compute(dimensions))

    def spec(self, *dimensions):
        """lazy handle: .full is forward(*dimensions); .table() its relative-position form for the LDS attention kernel"""
        return BiasSpec(self, tuple(dimensions))

    def table(self, *dimensions):
        """(tab (heads, L) f32, pos_code (n,) int32, offset, min, max, run4): bias[h][i][j] == tab[h][pos_code[i] - pos_code[j] + offset].
        The bias depends on (i, j) only through the relative grid position (attention.py:257-272), so the (heads, n, n) matrix --
        10.6 MB at (9, 8, 8), re-read by every (sequence, head) of every attention launch -- collapses to prod(2 d - 1) = 3 825
        values per head.  Built ONCE per (weights, dims) by scattering the exact entries of the full matrix (bit-identical values)."""
        params = list(self.parameters())

        def build():
            dims = tuple(dimensions)
            full = self._compute(dims)                                    # (heads, n, n), not kept
            dev = full.device
            grids = torch.meshgrid(*[torch.arange(d, device=dev) for d in dims], indexing='ij')
            strides, acc = [], 1
            for d in reversed(dims):
                strides.insert(0, acc)
                acc *= 2 * d - 1
            code = sum(g.reshape(-1) * st for g, st in zip(grids, strides))                  # (n,) mixed-radix position code
            off = sum((d - 1) * st for d, st in zip(dims, strides))
            idx = (code[:, None] - code[None, :] + off).reshape(-1)
            tab = torch.zeros((full.shape[0], acc), device=dev, dtype=torch.float32)
            tab[:, idx] = full.reshape(full.shape[0], -1)
            lo, hi = (float(v) for v in torch.stack((full.min(), full.max())).tolist())     # one host sync per (weights, dims)
            return tab.contiguous(), code.to(torch.int32).contiguous(), int(off), lo, hi, dims[-1] % 4 == 0
        return _cache(self).get(('table', tuple(dimensions)), params, build)

    def _compute(self, dims):
        dev = self.net[0][0].weight.device
        n = 1
        for d in dims:
            n *= d
        first = self.net[0][0]
        D = first.weight.shape[0]
        rows = n * n
        h = torch.empty((rows, D), device=dev, dtype=torch.float32)
        L.cpb_input(f32c(first.weight), f32c(first.bias), h, dims, D)
        for layer in list(self.net)[1:-1]:
            lin = layer[0]
            nxt = torch.empty((rows, lin.weight.shape[0]), device=dev, dtype=torch.float32)
            L.gemm(L.F32, h, pack_linear_weight(lin.weight, L.F32), rows, lin.weight.shape[0], lin.weight.shape[1],
                   C=nxt, bias=f32c(lin.bias), act=L.ACT_LEAKY)
            h = nxt
        last = self.net[-1]
        heads = last.weight.shape[0]
        out = torch.empty((rows, heads), device=dev, dtype=torch.float32)
        L.gemm(L.F32, h, pack_linear_weight(last.weight, L.F32), rows, heads, last.weight.shape[1], C=out, bias=f32c(last.bias))
        return out.reshape(n, n, heads).permute(2, 0, 1).contiguous()


# PK_QKV_ATTN=0: keep pk_qkv_project + pk_attn_fwd for short sequences too (A/B timing of the fused kernel)
_SHORT_FUSED = os.environ.get('PK_QKV_ATTN', '1') != '0'
# PK_CROSS_FUSED=0: cross-attention against the cached context keeps pk_qkv_project + pk_attn_fwd (A/B timing of pk_q_attn_cached)
_CROSS_FUSED = os.environ.get('PK_CROSS_FUSED', '1') != '0'
# PK_BIAS_TABLE=0: the n >= 64 attention kernel streams the full (heads, n, n) position bias instead of its relative-position table
_BIAS_TABLE = os.environ.get('PK_BIAS_TABLE', '1') != '0' and os.environ.get('PK_ATTN_LDS', '1') != '0'
# PK_CFG_SHARED_PREFIX=0: the cond | null halves of a CFG batch run layer 0's PEG + self-attention separately (they are identical there)
_CFG_SHARED_PREFIX = os.environ.get('PK_CFG_SHARED_PREFIX', '1') != '0'
# PK_ATTN_FIXED=0: the n > 64 attention kernel keeps a running maximum (flash) instead of the fixed-offset softmax (pk_attn_fwd score_bound)
_ATTN_FIXED = os.environ.get('PK_ATTN_FIXED', '1') != '0'


class Attention(PackedModule):
    """attention.py:89-182."""

    def __init__(self, dim, dim_context=None, dim_head=64, heads=8, causal=False, num_null_kv=0,
                 norm_context=True, dropout=0., scale=8):
        super().__init__()
        assert dim_head == 64, 'the MI355X attention kernels are built for dim_head = 64 (the reference default)'
        self.heads = heads
        self.causal = causal
        self.scale = scale
        inner_dim = dim_head * heads
        dim_context = default(dim_context, dim)
        if causal:
            self.rel_pos_bias = AlibiPositionalBias(heads=heads)
        self.attn_dropout = nn.Dropout(dropout)
        self.norm = LayerNorm(dim)
        self.context_norm = LayerNorm(dim_context) if norm_context else nn.Identity()
        self.num_null_kv = num_null_kv
        self.null_kv = nn.Parameter(torch.randn(heads, 2 * num_null_kv, dim_head))
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim_context, inner_dim * 2, bias=False)
        self.q_scale = nn.Parameter(torch.ones(dim_head))
        self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.to_out = nn.Linear(inner_dim, dim, bias=False)

    # -- key/value side: step-invariant for cross-attention, so callers may cache the result
    def project_kv(self, kv_src2d, S, n_kv, dtype, is_context):
        """kv_src2d: (S*n_kv, dim_kv) f32.  Self-attention passes the UN-normalised x (attention.py:140-144);
        cross-attention passes the raw context, normalised here by context_norm."""
        dev = kv_src2d.device
        td = L.tdtype(dtype)
        inner = self.to_q.weight.shape[0]
        Mk, Dk = kv_src2d.shape
        if is_context and isinstance(self.context_norm, LayerNorm):
            src = torch.empty((Mk, Dk), device=dev, dtype=td)
            self.context_norm.run(kv_src2d, out=src)
        else:
            src = kv_src2d
        kv = torch.empty((Mk, 2 * inner), device=dev, dtype=torch.float32)
        L.gemm(dtype, src, linear_weight(self.to_kv, dtype), Mk, 2 * inner, Dk, C=kv)
        return kv

    def _finish(self, o, x2d, dtype, want_t, dup=1):
        """to_out projection + residual  [+ the bf16 copy of the result for the next block's folded LayerNorm; want_t == 'stats': and
        the row statistics of that copy for a folded feed-forward LayerNorm -> (out, out_t, stats)].  dup = 2: the result (and its bf16
        copy) is written twice, rows [0, M) and [M, 2M) -- the cond | null copies of a CFG batch that was identical up to here."""
        M, D = x2d.shape
        out = torch.empty((dup * M, D), device=x2d.device, dtype=torch.float32)
        out_t = torch.empty((dup * M, D), device=x2d.device, dtype=torch.bfloat16) if (want_t and dtype == L.BF16) else None
        stats = None
        if want_t == 'stats' and (out_t is not None or dtype == L.BF16X3) and D % 4 == 0 and dup == 1:
            # (split-bf16: the statistics of the f32 rows themselves, which is what its folded feed-forward GEMM reads)
            stats = torch.empty((M, (D + 31) // 32, 2), device=x2d.device, dtype=torch.float32)
        L.gemm(dtype, o, linear_weight(self.to_out, dtype), M, D, o.shape[1], C=out, res=x2d, C2=out_t, stats_out=stats,
               dup_rows=M if dup == 2 else 0)
        if want_t == 'stats':
            return out, out_t, stats
        return (out, out_t) if want_t else out

    def _folded_q(self, dtype):
        """to_q with this block's LayerNorm folded in (gamma (.) Wq, s, t); None when the (zero) beta buffer is not zero"""
        wg, s, t, beta_zero = folded_weight(self.to_q, 'q_ln', lambda: self.to_q.weight.detach(), self.norm.gamma, self.norm.beta, dtype,
                                            [self.to_q.weight, self.norm.gamma, self.norm.beta])
        return (wg, s, t) if beta_zero else None

    def run(self, x2d, S, n, dtype, *, context2d=None, n_ctx=None, attn_bias=None, kmask=None, kv_cache=None, xt=None, want_t=False, dup=1):
        """x2d (S*n, D) f32 -> attention(x) + x.  kmask: (S, n_kv) uint8/bool over the real keys or None.
        xt: the bf16 copy of x2d its producer wrote (bf16 mode: the block's LayerNorm is folded into to_q, K / V read the same
        un-normalised rows); want_t: return (out, bf16 copy of out).  dup = 2: the outputs hold the result twice (see _finish)."""
        dev = x2d.device
        td = L.tdtype(dtype)
        M, D = x2d.shape
        h = self.heads
        inner = self.to_q.weight.shape[0]
        nnull = self.num_null_kv
        is_cross = context2d is not None
        n_kv = n_ctx if is_cross else n
        slopes = self.rel_pos_bias.slopes if self.causal else None
        cached = kv_cache.get(id(self)) if (kv_cache is not None and is_cross) else None
        # a position bias given as a BiasSpec reaches the bf16 LDS attention kernel (n >= 64 keys and queries, no null keys / mask /
        # causal) as a 15 KB relative-position table instead of the (heads, n, n) matrix; every other consumer takes the matrix
        bias_table = None
        spec = attn_bias if isinstance(attn_bias, BiasSpec) else None
        x3_lds = dtype == L.BF16X3 and n >= 128                 # split-bf16: the LDS-staged kernel exists in its fixed-offset form only (128-row workgroups)
        if (isinstance(attn_bias, BiasSpec) and _BIAS_TABLE and (dtype == L.BF16 or x3_lds) and not is_cross and nnull == 0 and n >= 64 and
                not (_SHORT_FUSED and n <= 64) and kmask is None and not self.causal):
            bias_table, attn_bias = attn_bias.table(), None
        attn_bias = _full_bias(attn_bias)
        # fixed-offset softmax for the long self-attention (pk_attn_fwd score_bound): q^ / k^ are unit vectors times q_scale / k_scale,
        # so |sim| <= scale * max|q_scale . k_scale| (+ 1/64 for the bf16 rounding of the operand images); the table knows its extremes
        score_bound = None
        if (_ATTN_FIXED and (dtype == L.BF16 or x3_lds) and not is_cross and nnull == 0 and n > 64 and kmask is None and not self.causal and
                (attn_bias is None)):
            c = _cache(self).get(('qk_bound',), [self.q_scale, self.k_scale],
                                 lambda: float((self.q_scale.detach().float() * self.k_scale.detach().float()).abs().max()))
            qk = float(self.scale) * c * (1 + 1 / 64)
            lo, hi = (bias_table[3], bias_table[4]) if bias_table is not None else (0., 0.)
            if math.isfinite(qk + hi - lo) and (2 * qk + hi - lo) * 1.4427 < 64:       # else: exponent range too wide for one offset
                score_bound = qk + hi
        if dtype == L.BF16X3 and bias_table is not None and score_bound is None:
            bias_table, attn_bias = None, spec.full            # no single exponent offset covers the range: the matrix form on the running-max kernel

        fq = self._folded_q(dtype) if ln_fold_enabled(dtype) else None
        if fq is not None:
            # ---- bf16, LayerNorm folded: q = l2norm(x (gamma.Wq)^T - mean(x) s)  (the l2norm cancels rstd), K / V from the same x
            wq, sq, tq = fq
            if xt is None:
                xt = x2d.to(td)
            if not is_cross and nnull == 0 and _SHORT_FUSED and n <= 64 and kmask is None and (attn_bias is None or attn_bias.stride(-1) == 1):
                # short sequences (C-ViViT spatial n = 64 / temporal n = 9..10): projections + attention in ONE launch
                o = torch.empty((M, inner), device=dev, dtype=td)
                L.qkv_attn(xt, xt, wq, linear_weight(self.to_kv, dtype), S, n, h, D, self.q_scale, self.k_scale, float(self.scale), o,
                           bias=attn_bias, slopes=slopes, causal=self.causal, q_ln_s=sq)
                return self._finish(o, x2d, dtype, want_t, dup)
            nq_pad, nk_pad = L.attn_pads(n, n_kv, nnull)
            Qp = torch.empty((S * h * nq_pad * 64,), device=dev, dtype=td)
            if not is_cross and nnull == 0:
                Kp = torch.empty((S * h * nk_pad * 64,), device=dev, dtype=td)
                Vt = torch.empty((S * h * nk_pad * 64,), device=dev, dtype=td)      # pad columns: masked inside the attention kernels
                L.qkv_project(xt, xt, wq, linear_weight(self.to_kv, dtype), S, n, h, D, self.q_scale, self.k_scale, float(self.scale),
                              Qp, Kp, Vt, nq_pad, nk_pad, q_ln_s=sq)
            elif cached is not None:
                Kp, Vt = cached                       # step-invariant context: only the query side is projected again
                if _CROSS_FUSED and n % 64 == 0 and nnull + n_kv <= 64:
                    # few keys (the text context): query projection + attention against the cached images in ONE launch
                    o = torch.empty((M, inner), device=dev, dtype=td)
                    L.q_attn_cached(xt, wq, S, n, h, D, self.q_scale, float(self.scale), Kp, Vt, nk_pad, n_kv, nnull, o, kmask=kmask, q_ln_s=sq)
                    return self._finish(o, x2d, dtype, want_t, dup)
                L.qkv_project(xt, None, wq, None, S, n, h, D, self.q_scale, None, float(self.scale), Qp, None, None, nq_pad, nk_pad, q_ln_s=sq)
            else:
                q = torch.empty((M, inner), device=dev, dtype=torch.float32)
                L.gemm(dtype, xt, wq, M, inner, D, C=q, ln=(sq, tq, self.norm.eps))
                kv = self.project_kv(context2d if is_cross else xt, S, n_kv, dtype, is_cross)
                Kp = torch.empty((S * h * nk_pad * 64,), device=dev, dtype=td)
                Vt = torch.empty((S * h * nk_pad * 64,), device=dev, dtype=td)
                L.attn_prep(dtype, q, kv, self.null_kv, self.q_scale, self.k_scale, float(self.scale), Qp, Kp, Vt, S, h, n, n_kv, nnull)
                if kv_cache is not None and is_cross:
                    kv_cache[id(self)] = (Kp, Vt)
            o = torch.empty((M, inner), device=dev, dtype=td)
            L.attn_fwd(dtype, Qp, Kp, Vt, o, S, h, n, n_kv, nnull, bias=attn_bias, kmask=kmask, slopes=slopes, causal=self.causal,
                       bias_table=bias_table, score_bound=score_bound)
            return self._finish(o, x2d, dtype, want_t, dup)

        # ---- separate LayerNorm launch (exact-f32 mode; bf16 with PK_LN_FOLD=0)
        xn = torch.empty((M, D), device=dev, dtype=td)
        # self-attention K/V come from the UN-normalised x (attention.py:140-144): in bf16 mode the same LN launch
        # also emits x in bf16 so every GEMM operand is T and can be fed by LDS-DMA
        xraw = torch.empty((M, D), device=dev, dtype=td) if (not is_cross and dtype == L.BF16) else None
        L.layernorm(x2d, self.norm.gamma, self.norm.beta, M, D, out=xn, raw=xraw)
        # exact-f32 mode, very short sequences: one fused f32 launch; in bf16 mode the fused projection + MFMA attention
        # measured faster for the temporal layers (1.030 vs 1.046 ms per encode step)
        small = not is_cross and nnull == 0 and n <= 16 and dtype != L.BF16
        # bf16: to_q / to_kv run as ONE GEMM launch whose epilogue writes the attention operand images directly
        # (pk_qkv_project); available for self-attention (no null keys) and for cross-attention with cached K / V
        fused = dtype == L.BF16 and not small and ((not is_cross and nnull == 0) or cached is not None)

        if _SHORT_FUSED and fused and not is_cross and n <= 64 and kmask is None and (attn_bias is None or attn_bias.stride(-1) == 1):
            o = torch.empty((M, inner), device=dev, dtype=td)
            L.qkv_attn(xn, xraw, linear_weight(self.to_q, dtype), linear_weight(self.to_kv, dtype), S, n, h, D, self.q_scale,
                       self.k_scale, float(self.scale), o, bias=attn_bias, slopes=slopes, causal=self.causal)
            return self._finish(o, x2d, dtype, want_t, dup)

        q = None
        if not fused:
            q = torch.empty((M, inner), device=dev, dtype=torch.float32)
            L.gemm(dtype, xn, linear_weight(self.to_q, dtype), M, inner, D, C=q)

        if small:
            # very short sequences (C-ViViT temporal layers, n = 9..10): l2norm, scales, ALiBi, softmax and PV in ONE
            # launch straight from the projection outputs (measured: 21 us vs 26 us for prep + MFMA attention; at n = 64
            # the f32 VALU loop loses to the MFMA path, 59 us vs 25 us, so the spatial layers keep that)
            kv = self.project_kv(xraw if xraw is not None else x2d, S, n_kv, dtype, False)
            o = torch.empty((M, inner), device=dev, dtype=td)
            L.attn_small(q, kv, self.q_scale, self.k_scale, float(self.scale), o, S, h, n, bias=attn_bias, kmask=kmask,
                         slopes=slopes, causal=self.causal)
            return self._finish(o, x2d, dtype, want_t, dup)

        nq_pad, nk_pad = L.attn_pads(n, n_kv, nnull)
        Qp = torch.empty((S * h * nq_pad * 64,), device=dev, dtype=td)
        if fused:
            if is_cross:
                Kp, Vt = cached
                L.qkv_project(xn, None, linear_weight(self.to_q, dtype), None, S, n, h, D, self.q_scale, None, float(self.scale),
                              Qp, None, None, nq_pad, nk_pad)
            else:
                Kp = torch.empty((S * h * nk_pad * 64,), device=dev, dtype=td)
                # V^T pad columns (keys >= n) are never written: the attention kernels mask them in the tail tile
                Vt = torch.empty((S * h * nk_pad * 64,), device=dev, dtype=td)
                L.qkv_project(xn, xraw, linear_weight(self.to_q, dtype), linear_weight(self.to_kv, dtype), S, n, h, D,
                              self.q_scale, self.k_scale, float(self.scale), Qp, Kp, Vt, nq_pad, nk_pad)
        elif cached is None:
            kv = self.project_kv(context2d if is_cross else (xraw if xraw is not None else x2d), S, n_kv, dtype, is_cross)
            Kp = torch.empty((S * h * nk_pad * 64,), device=dev, dtype=td)
            Vt = torch.empty((S * h * nk_pad * 64,), device=dev, dtype=td)
            L.attn_prep(dtype, q, kv, self.null_kv, self.q_scale, self.k_scale, float(self.scale), Qp, Kp, Vt, S, h, n, n_kv, nnull)
            if kv_cache is not None and is_cross:
                kv_cache[id(self)] = (Kp, Vt)
        else:
            Kp, Vt = cached      # step-invariant context: only the query side is prepared again
            L.attn_prep(dtype, q, None, self.null_kv, self.q_scale, self.k_scale, float(self.scale), Qp, None, None, S, h, n, n_kv, nnull)
        o = torch.empty((M, inner), device=dev, dtype=td)
        L.attn_fwd(dtype, Qp, Kp, Vt, o, S, h, n, n_kv, nnull, bias=attn_bias, kmask=kmask, slopes=slopes, causal=self.causal,
                   bias_table=bias_table, score_bound=score_bound)
        return self._finish(o, x2d, dtype, want_t, dup)

    def forward(self, x, mask=None, context=None, attn_bias=None):
        L.require_device(x, 'x')
        S, n, D = x.shape
        x2 = x.reshape(S * n, D).float().contiguous()
        ctx2, n_ctx = None, None
        if exists(context):
            n_ctx = context.shape[1]
            ctx2 = context.reshape(S * n_ctx, context.shape[-1]).float().contiguous()
        km = mask.to(torch.uint8).contiguous() if exists(mask) else None
        ab = attn_bias.float().contiguous() if exists(attn_bias) else None
        out = self.run(x2, S, n, compute_dtype_of(self), context2d=ctx2, n_ctx=n_ctx, attn_bias=ab, kmask=km)
        return (out - x2).reshape(x.shape)


def _unpack(r):
    """out | (out, out_t) | (out, out_t, stats) -> (out, out_t, stats)"""
    if not isinstance(r, tuple):
        return r, None, None
    return r if len(r) == 3 else (r[0], r[1], None)


class Transformer(PackedModule):
    """attention.py:279-332 : per layer [PEG?, self Attention, cross Attention?, FeedForward], each + residual; norm_out."""

    def shares_cfg_prefix(self, dtype, context2d, self_attn_mask):
        """True when run(..., replicas=2) may be used: bf16 with folded LayerNorms (the path that threads the bf16 copy), layer 0 has a
        cross-attention that will run, and no per-sequence self-attention mask (the two copies must be identical up to there)."""
        if not (_CFG_SHARED_PREFIX and ln_fold_enabled(dtype) and len(self.layers) > 0):
            return False
        cross = self.layers[0][2]
        return exists(cross) and exists(context2d) and self_attn_mask is None

    def __init__(self, dim, *, depth, dim_context=None, causal=False, dim_head=64, heads=8, ff_mult=4, peg=False,
                 peg_causal=False, attn_num_null_kv=2, has_cross_attn=False, attn_dropout=0., ff_dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PEG(dim=dim, causal=peg_causal) if peg else None,
                Attention(dim=dim, dim_head=dim_head, heads=heads, causal=causal, dropout=attn_dropout),
                Attention(dim=dim, dim_head=dim_head, dim_context=dim_context, heads=heads, causal=False,
                          num_null_kv=attn_num_null_kv, dropout=attn_dropout) if has_cross_attn else None,
                FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout),
            ]))
        self.norm_out = LayerNorm(dim)

    def run(self, x2d, S, n, dtype, *, video_shape=None, attn_bias=None, context2d=None, n_ctx=None,
            self_attn_mask=None, cross_attn_context_mask=None, kv_cache=None, out=None, out_t=None, perm=(0, 0),
            skip_norm_out=False, xt=None, replicas=1):
        """x2d (S*n, D) f32.  Writes norm_out(x) to `out` (f32) and/or `out_t` (T); returns `out` (allocated if both None).
        perm = (pb, pc): the output rows are written transposed, (a, b, c) -> (a, c, b).
        xt: the bf16 copy of x2d, if the producer already wrote one (bf16 mode; else the first block converts).
        skip_norm_out: return the residual stream BEFORE norm_out (the caller fuses that LayerNorm into its next kernel).
        replicas = 2 (see shares_cfg_prefix): the S sequences are the cond | null copies of S / 2 sequences, identical until the first
        cross-attention; x2d holds the S / 2 distinct ones, layer 0's PEG + self-attention run once and write both copies."""
        x = x2d
        fold = ln_fold_enabled(dtype)
        nl = len(self.layers)
        assert replicas == 1 or self.shares_cfg_prefix(dtype, context2d, self_attn_mask), 'replicas: see shares_cfg_prefix'
        S_cur = S // replicas
        for li, (peg, self_attn, cross_attn, ff) in enumerate(self.layers):
            if fold:
                # bf16: a block whose LayerNorm is folded into its first GEMM reads the bf16 copy (xt) of the residual stream, which the
                # block BEFORE it writes beside the f32 one (only when somebody will read it)
                has_cross = exists(cross_attn) and exists(context2d)
                if exists(peg):
                    x, xt = _unpack(peg.run(x, video_shape if S_cur == S else (video_shape[0] * S_cur // S, *video_shape[1:]),
                                            want_t=dtype == L.BF16))[:2]           # (split-bf16 reads the f32 rows: no copy)
                ff_wants = {0: False, 1: True, 2: 'stats'}[ff_fold_mode(S * n)]           # what the block in front of the FF leaves for it
                x, xt, stats = _unpack(self_attn.run(x, S_cur, n, dtype, attn_bias=attn_bias, kmask=self_attn_mask, xt=xt,
                                                     want_t=True if has_cross else ff_wants, dup=S // S_cur))
                S_cur = S
                if has_cross:
                    x, xt, stats = _unpack(cross_attn.run(x, S, n, dtype, context2d=context2d, n_ctx=n_ctx, kmask=cross_attn_context_mask,
                                                          kv_cache=kv_cache, xt=xt, want_t=ff_wants))
                next_reads_xt = li + 1 < nl and not exists(self.layers[li + 1][0])       # next layer starts with attention (no PEG)
                x, xt, _ = _unpack(ff.run(x, dtype, xt=xt, want_t=next_reads_xt, stats=stats))
                continue
            if exists(peg):
                x = peg.run(x, video_shape)
            x = self_attn.run(x, S, n, dtype, attn_bias=attn_bias, kmask=self_attn_mask)
            if exists(cross_attn) and exists(context2d):
                x = cross_attn.run(x, S, n, dtype, context2d=context2d, n_ctx=n_ctx, kmask=cross_attn_context_mask,
                                   kv_cache=kv_cache)
            x = ff.run(x, dtype)
        if skip_norm_out:
            return x
        if out is None and out_t is None:
            out = torch.empty_like(x)
        M, D = x.shape
        L.layernorm(x, self.norm_out.gamma, self.norm_out.beta, M, D, out=out_t, out2=out, perm=perm)
        return out if out is not None else out_t

    def forward(self, x, video_shape=None, attn_bias=None, context=None, self_attn_mask=None,
                cross_attn_context_mask=None, _kv_cache=None):
        L.require_device(x, 'x')
        S, n, D = x.shape
        x2 = x.reshape(S * n, D).float().contiguous()
        ctx2, n_ctx = None, None
        if exists(context):
            n_ctx = context.shape[1]
            ctx2 = context.reshape(S * n_ctx, context.shape[-1]).float().contiguous()
        sm = self_attn_mask.to(torch.uint8).contiguous() if exists(self_attn_mask) else None
        cm = cross_attn_context_mask.to(torch.uint8).contiguous() if exists(cross_attn_context_mask) else None
        ab = attn_bias.float().contiguous() if exists(attn_bias) else None
        out = self.run(x2, S, n, compute_dtype_of(self), video_shape=tuple(video_shape) if exists(video_shape) else None,
                       attn_bias=ab, context2d=ctx2, n_ctx=n_ctx, self_attn_mask=sm, cross_attn_context_mask=cm,
                       kv_cache=_kv_cache)
        return out.reshape(S, n, D)
