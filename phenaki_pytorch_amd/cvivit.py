"""C-ViViT video tokenizer, inference surface of /root/reference/phenaki_pytorch/cvivit.py:226-583, on MI355X kernels.

Same constructor signature, attribute names and state_dict keys as the reference `CViViT`; `forward(...,
return_only_codebook_ids=True)`, `forward(..., return_recons_only=True)`, `encode`, `decode`,
`decode_from_codebook_indices` and the shape helpers run here; with `use_vgg_and_gan=False` the default `forward(video)`
returns the VALUE of the reconstruction loss (cvivit.py:585-627, no autograd graph) under no_grad and the training step's loss
with a graph otherwise (train_cvivit.py).  `use_vgg_and_gan=True` builds the Discriminator (discriminator.py); the perceptual /
adversarial / adaptive-weight objective and `return_discr_loss=True` (cvivit.py:604-671) run in train_cvivit.py.
"""
import contextlib
import copy
import os
from pathlib import Path

import torch
from torch import nn

from . import _lib as L
from .attention import (ContinuousPositionBias, PackedModule, Transformer, compute_dtype_of, exists, folded_weight, invalidate_packed,
                        linear_weight, ln_fold_enabled, value_without_graph, set_compute_dtype)
from .quantize import LFQ, VectorQuantize

# PK_PATCH_FUSED=0: the bf16 patch embedding keeps pk_patchify_ln + pk_gemm (A/B timing of the fused pk_patch_embed)
_PATCH_FUSED = os.environ.get('PK_PATCH_FUSED', '1') != '0'
# PK_PATCH_WIDE=0: the round-3 tiling of the fused patch embedding (128 x 128 tiles, pk_patch_embed + pk_layernorm) for A/B timing
_PATCH_WIDE = os.environ.get('PK_PATCH_WIDE', '1') != '0'
# both frame groups' pk_patch_embed_finish in one launch (0: one launch per group, the round-4 path, for A/B timing)
_PATCH_FINISH_ONE = os.environ.get('PK_PATCH_FINISH_ONE', '1') != '0'
_PE_SPLITK = int(os.environ.get('PK_PE_SPLITK', '4'))          # K-slices of the split-bf16 patch-embedding GEMM (1: one plain launch)
_PE_SPLITK_FIRST = int(os.environ.get('PK_PE_SPLITK_FIRST', '8'))   # ... of its first-frame group (few rows, 64 x 64 tiles)


def pair(val):
    ret = (val, val) if not isinstance(val, tuple) else val
    assert len(ret) == 2
    return ret


def divisible_by(numer, denom):
    return (numer % denom) == 0


def _pe_bias(lin, t):
    """t + the Linear's bias (cached with the folded weight's lifetime: rebuilt when either parameter changes)"""
    from .attention import _cache
    if lin.bias is None:
        return t
    return _cache(lin).get(('pe_t',), [lin.weight, lin.bias, t], lambda: (t + lin.bias.detach().float()).contiguous())


class _Rearrange(nn.Identity):
    """placeholder keeping the reference's nn.Sequential indices (the einops Rearrange has no parameters;
    the layout change is fused into pk_patchify_ln / pk_unpatchify)."""


class CViViT(PackedModule):
    def __init__(self, *, dim, codebook_size, image_size, patch_size, temporal_patch_size, spatial_depth,
                 temporal_depth, discr_base_dim=16, dim_head=64, heads=8, channels=3, use_vgg_and_gan=True,
                 vgg=None, discr_attn_res_layers=(16,), use_hinge_loss=True, attn_dropout=0., ff_dropout=0.,
                 lookup_free_quantization=True, lookup_free_quantization_kwargs: dict = {}):
        super().__init__()
        self.image_size = pair(image_size)
        self.patch_size = pair(patch_size)
        patch_height, patch_width = self.patch_size
        self.temporal_patch_size = temporal_patch_size
        self.channels = channels
        self.dim = dim

        self.spatial_rel_pos_bias = ContinuousPositionBias(dim=dim, heads=heads)

        image_height, image_width = self.image_size
        assert (image_height % patch_height) == 0 and (image_width % patch_width) == 0

        p1 = channels * patch_width * patch_height
        p2 = p1 * temporal_patch_size
        self.to_patch_emb_first_frame = nn.Sequential(_Rearrange(), nn.LayerNorm(p1), nn.Linear(p1, dim), nn.LayerNorm(dim))
        self.to_patch_emb = nn.Sequential(_Rearrange(), nn.LayerNorm(p2), nn.Linear(p2, dim), nn.LayerNorm(dim))

        spatial_kwargs = dict(dim=dim, dim_head=dim_head, heads=heads, attn_dropout=attn_dropout,
                              ff_dropout=ff_dropout, causal=False, peg=False)
        temporal_kwargs = dict(dim=dim, dim_head=dim_head, heads=heads, attn_dropout=attn_dropout,
                               ff_dropout=ff_dropout, causal=True, peg=True, peg_causal=True)

        self.enc_spatial_transformer = Transformer(depth=spatial_depth, **spatial_kwargs)
        self.enc_temporal_transformer = Transformer(depth=temporal_depth, **temporal_kwargs)

        self.lookup_free_quantization = lookup_free_quantization
        if lookup_free_quantization:
            self.vq = LFQ(dim=dim, codebook_size=codebook_size, **lookup_free_quantization_kwargs)
        else:
            self.vq = VectorQuantize(dim=dim, codebook_size=codebook_size, use_cosine_sim=True)

        self.dec_spatial_transformer = Transformer(depth=spatial_depth, **spatial_kwargs)
        self.dec_temporal_transformer = Transformer(depth=temporal_depth, **temporal_kwargs)

        self.to_pixels_first_frame = nn.Sequential(nn.Linear(dim, p1), _Rearrange())
        self.to_pixels = nn.Sequential(nn.Linear(dim, p2), _Rearrange())

        # VQGAN training branch (cvivit.py:336-363): not on the inference hot path; the tokenizer's own training uses it (train_cvivit.py)
        self.vgg = None
        self.discr = None
        self.use_vgg_and_gan = use_vgg_and_gan
        self.use_hinge_loss = use_hinge_loss
        if not use_vgg_and_gan:
            return
        if exists(vgg):
            self.vgg = vgg
        else:
            # the reference default is torchvision's pretrained VGG16 with the last two classifier layers cut (cvivit.py:349-352); it needs
            # torchvision and a download.  Without them the module still builds (inference and the discriminator step need no VGG); the
            # generator's GAN step then asks for `vgg=` (train_cvivit.py).
            try:
                import torchvision
                self.vgg = torchvision.models.vgg16(pretrained=True)
                self.vgg.classifier = nn.Sequential(*self.vgg.classifier[:-2])
            except Exception as e:                                    # noqa: BLE001  (ImportError offline, URLError without network, ...)
                import warnings
                warnings.warn(f'CViViT(use_vgg_and_gan=True): no perceptual network ({type(e).__name__}: {e}); pass vgg=<nn.Module> to train '
                              'with the perceptual + adversarial losses')
        from .discriminator import Discriminator
        self.discr = Discriminator(image_size=self.image_size, dim=discr_base_dim, channels=channels, attn_res_layers=discr_attn_res_layers)

    # ---------------------------------------------------------------- host-side helpers (cvivit.py:365-447)

    def set_compute_dtype(self, name):
        return set_compute_dtype(self, name)

    def calculate_video_token_mask(self, videos, video_frame_mask):
        *_, h, w = videos.shape
        ph, pw = self.patch_size
        assert torch.all(((video_frame_mask.sum(dim=-1) - 1) % self.temporal_patch_size) == 0), \
            'number of frames must be divisible by temporal patch size, subtracting off the first frame'
        first, rest = video_frame_mask[:, :1], video_frame_mask[:, 1:]
        rest = rest.reshape(rest.shape[0], -1, self.temporal_patch_size)
        video_mask = torch.cat((first, rest.any(dim=-1)), dim=-1)
        return video_mask.repeat_interleave((h // ph) * (w // pw), dim=-1)

    def get_video_patch_shape(self, num_frames, include_first_frame=True):
        patch_frames = 0
        if include_first_frame:
            num_frames -= 1
            patch_frames += 1
        patch_frames += (num_frames // self.temporal_patch_size)
        return (patch_frames, *self.patch_height_width)

    @property
    def image_num_tokens(self):
        return int(self.image_size[0] / self.patch_size[0]) * int(self.image_size[1] / self.patch_size[1])

    def frames_per_num_tokens(self, num_tokens):
        tokens_per_frame = self.image_num_tokens
        assert (num_tokens % tokens_per_frame) == 0, f'number of tokens must be divisible by number of tokens per frame {tokens_per_frame}'
        assert (num_tokens > 0)
        pseudo_frames = num_tokens // tokens_per_frame
        return (pseudo_frames - 1) * self.temporal_patch_size + 1

    def num_tokens_per_frames(self, num_frames, include_first_frame=True):
        image_num_tokens = self.image_num_tokens
        total_tokens = 0
        if include_first_frame:
            num_frames -= 1
            total_tokens += image_num_tokens
        assert (num_frames % self.temporal_patch_size) == 0
        return total_tokens + int(num_frames / self.temporal_patch_size) * image_num_tokens

    def copy_for_eval(self):
        device = next(self.parameters()).device
        invalidate_packed(self)                       # packed device weights are rebuilt lazily, never copied
        vae_copy = copy.deepcopy(self)
        if vae_copy.use_vgg_and_gan:                  # cvivit.py:415-417 (`del`; here the attributes stay, as None: forward() reads them)
            vae_copy.discr = None
            vae_copy.vgg = None
        vae_copy.eval()
        return vae_copy.to(device)

    @contextlib.contextmanager
    def _without_vgg(self):
        """@remove_vgg (cvivit.py:35-49): the perceptual network is never part of a checkpoint"""
        vgg = self._modules.get('vgg')
        if isinstance(vgg, nn.Module):
            del self._modules['vgg']
        try:
            yield
        finally:
            if isinstance(vgg, nn.Module):
                self._modules['vgg'] = vgg

    def state_dict(self, *args, **kwargs):
        with self._without_vgg():
            return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        # cvivit.py:424-426; a checkpoint of a GAN-trained tokenizer also loads into a module built without the discriminator
        sd = {k: v for k, v in state_dict.items() if not (k.startswith('vgg.') or (self.discr is None and k.startswith('discr.')))}
        with self._without_vgg():
            return super().load_state_dict(sd, *args, **kwargs)        # PackedModule: also drops the packed weight caches

    def load(self, path):
        path = Path(path)
        assert path.exists()
        pt = torch.load(str(path))
        self.load_state_dict(pt)

    @property
    def patch_height_width(self):
        return self.image_size[0] // self.patch_size[0], self.image_size[1] // self.patch_size[1]

    # ---------------------------------------------------------------- kernels-backed pieces (2-D f32 token buffers)

    def _patch_embed(self, video):
        """(B,C,F,H,W) f32 -> tokens (B*T'*h*w, dim) f32 in (b t h w) row order (cvivit.py:542-549)."""
        dt = compute_dtype_of(self)
        td = L.tdtype(dt)
        B, C, F, H, W = video.shape
        ph, pw = self.patch_size
        h, w = self.patch_height_width
        hw = h * w
        pt = self.temporal_patch_size
        nt = (F - 1) // pt
        T = 1 + nt
        dev = video.device
        tokens = torch.empty((B * T * hw, self.dim), device=dev, dtype=torch.float32)
        # bf16 copy for the first transformer block (its LayerNorm is folded into its first GEMM, which reads bf16 rows)
        tokens_t = torch.empty((B * T * hw, self.dim), device=dev, dtype=td) if (ln_fold_enabled(dt) and dt == L.BF16) else None

        groups = [(self.to_patch_emb, 1, nt, pt, hw)] if nt > 0 else []
        groups.append((self.to_patch_emb_first_frame, 0, 1, 1, 0))                     # long-K group first
        fused = (_PATCH_FUSED and dt == L.BF16 and pw in (8, 16, 32, 64, 128) and all((C * ptg * ph * pw) % 192 == 0 for _, _, _, ptg, _ in groups)
                 and (pw >= 32 or ph % (32 // pw) == 0) and self.dim % 4 == 0 and video.numel() * 4 < 0xFFFFFFF0 and
                 all(seq[1].eps == groups[0][0][1].eps for seq, *_ in groups))
        if fused and _PATCH_WIDE and self.dim <= 512:
            # round 4: row panels x all columns x K-slices (pk_patch_embed_splitk: the video read once, 204 workgroups at B = 8), then per group
            # ONE finish launch = slice sum + folded LayerNorm(P) + bias + the LayerNorm(dim) that follows
            spec, fin = [], []
            for seq, f0, ntg, ptg, goff in groups:
                ln1, lin, ln2 = seq[1], seq[2], seq[3]
                P = C * ptg * ph * pw
                wg, s_, t_, _ = folded_weight(lin, 'pe_ln', lambda lin=lin: lin.weight.detach(), ln1.weight, ln1.bias, dt, [lin.weight, ln1.weight, ln1.bias])
                ns, rows = L.patch_embed_slices(P), B * ntg * hw
                part = torch.empty((ns, rows, self.dim), device=dev, dtype=torch.float32)
                stats = torch.empty((ns, rows, 2), device=dev, dtype=torch.float32)
                spec.append((wg, part, stats, f0, ntg, ptg))
                fin.append((part, stats, P, s_, _pe_bias(lin, t_), ln1.eps, ln2, (ntg * hw, T * hw, goff)))
            L.patch_embed_splitk(video, ph, pw, self.dim, spec)
            if _PATCH_FINISH_ONE:
                # round 5: both groups' slice folding + LayerNorms in ONE launch (the first-frame group is 512 rows: its own launch was all ramp)
                L.patch_embed_finish_groups([(part, stats, P, s_, tb, eps1, ln2.weight, ln2.bias, ln2.eps, remap) for part, stats, P, s_, tb, eps1, ln2, remap in fin],
                                            out2=tokens, out=tokens_t)
            else:
                for part, stats, P, s_, tb, eps1, ln2, remap in fin:
                    L.patch_embed_finish(part, stats, P, s_, tb, eps1, ln2.weight, ln2.bias, ln2.eps, out2=tokens, out=tokens_t, remap=remap)
            self.__dict__['_pk_tokens_t'] = tokens_t
            return tokens, T
        if fused:
            # ONE launch: patch gather + LayerNorm(P) + Linear for both frame groups (pk_patch_embed), then the LayerNorm(dim) per group
            spec, tmps = [], []
            for seq, f0, ntg, ptg, goff in groups:
                ln1, lin = seq[1], seq[2]
                wg, s_, t_, _ = folded_weight(lin, 'pe_ln', lambda lin=lin: lin.weight.detach(), ln1.weight, ln1.bias, dt, [lin.weight, ln1.weight, ln1.bias])
                tb = _pe_bias(lin, t_)
                tmp = torch.empty((B * ntg * hw, self.dim), device=dev, dtype=torch.float32)
                spec.append((wg, s_, tb, tmp, f0, ntg, ptg))
                tmps.append(tmp)
            L.patch_embed(video, ph, pw, self.dim, spec, eps=groups[0][0][1].eps)
            for (seq, f0, ntg, ptg, goff), tmp in zip(groups, tmps):
                ln2 = seq[3]
                L.layernorm(tmp, ln2.weight, ln2.bias, tmp.shape[0], self.dim, out=tokens_t, out2=tokens, eps=ln2.eps, remap=(ntg * hw, T * hw, goff))
            self.__dict__['_pk_tokens_t'] = tokens_t
            return tokens, T

        def group(seq, f0, ntg, ptg, goff):
            ln1, lin, ln2 = seq[1], seq[2], seq[3]
            P = C * ptg * ph * pw
            rows = B * ntg * hw
            patches = torch.empty((rows, P), device=dev, dtype=td)
            L.patchify_ln(video, f0, ntg, ptg, ph, pw, ln1.weight, ln1.bias, patches, eps=ln1.eps)
            tmp = torch.empty((rows, self.dim), device=dev, dtype=torch.float32)
            # split-bf16, the long-K group (P = 6144, 4096 rows at B = 8): 128 x 128 tiles alone are 128 workgroups; K-slices bring every CU in
            # (pk_gemm_splitk tile = 1, bias on slice 0; slices added in index order: deterministic)
            splits, tile = 1, 1
            if dt == L.BF16X3 and self.dim % 4 == 0 and _PE_SPLITK > 1:
                if rows >= 1024 and P >= 4096 and P % (32 * _PE_SPLITK) == 0:
                    splits = _PE_SPLITK
                elif _PE_SPLITK_FIRST > 1 and P >= 2048 and P % (32 * _PE_SPLITK_FIRST) == 0:
                    # the first-frame group (512 rows at B = 8, P = 3072): 64 tiles of 64 x 64 walking 96 k-tiles each leave 3/4 of the CUs idle
                    # (51.7 us); K-slices of 384 make it 512 workgroups
                    splits, tile = _PE_SPLITK_FIRST, 0
            if splits > 1:
                part = torch.empty((splits, rows * self.dim), device=dev, dtype=torch.float32)
                L.gemm_splitk(dt, patches, linear_weight(lin, dt), rows, self.dim, P, splits, part, bias=lin.bias, tile=tile)
                L.sum_batch(part, splits, tmp, rows * self.dim)
            else:
                L.gemm(dt, patches, linear_weight(lin, dt), rows, self.dim, P, C=tmp, bias=lin.bias)
            L.layernorm(tmp, ln2.weight, ln2.bias, rows, self.dim, out=tokens_t, out2=tokens, eps=ln2.eps, remap=(ntg * hw, T * hw, goff))

        group(self.to_patch_emb_first_frame, 0, 1, 1, 0)
        if nt > 0:
            group(self.to_patch_emb, 1, nt, pt, hw)
        self.__dict__['_pk_tokens_t'] = tokens_t          # picked up by tokenize() / forward() for the encoder's first block
        return tokens, T

    def _spatial(self, transformer, x2d, B, T, to_temporal=False, as_t=False, xt=None):
        """rows '(b t) (h w)'.  to_temporal: the final norm_out writes its rows as '(b h w) t' (cvivit.py:468) so the
        temporal transformer needs no transpose pass; as_t: return the rows in the GEMM operand type."""
        h, w = self.patch_height_width
        dt = compute_dtype_of(self)
        bias = self.spatial_rel_pos_bias(h, w)
        out_t = torch.empty((x2d.shape[0], x2d.shape[1]), device=x2d.device, dtype=L.tdtype(dt)) if as_t else None
        return transformer.run(x2d, B * T, h * w, dt, video_shape=(B, T, h, w), attn_bias=bias, out_t=out_t,
                               perm=(T, h * w) if to_temporal else (0, 0), xt=xt)

    def _temporal(self, transformer, xt2d, B, T, skip_norm_out=False, want_t=False):
        """rows '(b h w) t' in, '(b t) (h w)' out (the final norm_out writes transposed, cvivit.py:472,496).
        NOTE: video_shape stays (b, t, h, w) although rows are ((b h w), t): the reference's PEG sees that
        scrambled view (cvivit.py:456,468-470) and so must we."""
        h, w = self.patch_height_width
        dt = compute_dtype_of(self)
        if want_t and ln_fold_enabled(dt) and dt == L.BF16:        # (split-bf16 folds too, but reads the f32 rows: no second copy)
            # the final norm_out writes its rows twice: f32 (residual stream of the next transformer) and bf16 (its first GEMM operand)
            out = torch.empty_like(xt2d)
            out_t = torch.empty(xt2d.shape, device=xt2d.device, dtype=L.tdtype(dt))
            transformer.run(xt2d, B * h * w, T, dt, video_shape=(B, T, h, w), perm=(h * w, T), out=out, out_t=out_t)
            return out, out_t
        out = transformer.run(xt2d, B * h * w, T, dt, video_shape=(B, T, h, w), perm=(h * w, T), skip_norm_out=skip_norm_out)
        return (out, None) if want_t else out

    def _encode2d(self, tokens2d, B, T, skip_norm_out=False, xt=None):
        x = self._spatial(self.enc_spatial_transformer, tokens2d, B, T, to_temporal=True, xt=xt)
        return self._temporal(self.enc_temporal_transformer, x, B, T, skip_norm_out=skip_norm_out)

    def _unpatch_maps(self, B, T, f0, ntg, ptg, row0, dev):
        """(token rows of the frame group, video offset of every row, video offset of every pixel column), cached per geometry."""
        cache = self.__dict__.setdefault('_pk_unpatch_maps', {})
        key = (B, T, f0, ntg, ptg, row0, str(dev))
        if key not in cache:
            h, w = self.patch_height_width
            ph, pw = self.patch_size
            C, hw = self.channels, h * w
            F = 1 + (T - 1) * self.temporal_patch_size
            H, W = self.image_size
            if pw % 4 or W % 4:
                # no kernel of this build moves pixels in other than 16-byte pieces (pk_patchify_ln / pk_unpatchify refuse too)
                raise ValueError('patch width and image width must be multiples of 4 (16-byte pixel accesses)')
            if B * C * F * H * W >= 2 ** 31:
                # beyond 32-bit scatter offsets: _decode2d takes the (rows, P) pixel matrix + pk_unpatchify path (64-bit addressing)
                cache[key] = None
                return None
            ar = lambda n: torch.arange(n, dtype=torch.int64)
            b, tt, hh, ww = torch.meshgrid(ar(B), ar(ntg), ar(h), ar(w), indexing='ij')
            idx = (b * (T * hw) + row0 + tt * hw + hh * w + ww).reshape(-1)
            row_off = (((b * C * F + f0 + tt * ptg) * H + hh * ph) * W + ww * pw).reshape(-1)
            c, d, y, x = torch.meshgrid(ar(C), ar(ptg), ar(ph), ar(pw), indexing='ij')
            col_off = (((c * F + d) * H + y) * W + x).reshape(-1)
            cache[key] = tuple(t.to(device=dev, dtype=torch.int32).contiguous() for t in (idx, row_off, col_off))
        return cache[key]

    def _decode2d(self, tokens2d, B, T, temporal_rows=False):
        """tokens (B*T*h*w, dim) f32 -> video (B, C, 1 + (T-1)*pt, H, W) f32 (cvivit.py:476-516).
        temporal_rows: the rows already are in the '(b h w) t' order of the temporal transformer (pk_lfq_decode wrote them so)."""
        dt = compute_dtype_of(self)
        hw0 = self.image_num_tokens
        D0 = tokens2d.shape[-1]
        if temporal_rows:
            xt = tokens2d
        else:
            xt = tokens2d.view(B, T, hw0, D0).transpose(1, 2).contiguous().view(B * hw0 * T, D0)    # 'b t h w d -> (b h w) t d'
        x, x_t = self._temporal(self.dec_temporal_transformer, xt, B, T, want_t=True)
        x = self._spatial(self.dec_spatial_transformer, x, B, T, as_t=True, xt=x_t)
        h, w = self.patch_height_width
        hw = h * w
        ph, pw = self.patch_size
        pt = self.temporal_patch_size
        C = self.channels
        dev = x.device
        F = 1 + (T - 1) * pt
        H, W = self.image_size
        video = torch.empty((B, C, F, H, W), device=dev, dtype=torch.float32)
        for seq, f0, ntg, ptg, row0 in ((self.to_pixels_first_frame, 0, 1, 1, 0), (self.to_pixels, 1, T - 1, pt, hw)):
            if ntg <= 0:
                continue
            lin = seq[0]
            maps = self._unpatch_maps(B, T, f0, ntg, ptg, row0, dev)
            if maps is None:
                # geometry outside the scatter epilogue's constraints: (rows, P) pixel matrix, then the un-patchify pass
                rows, P = B * ntg * hw, C * ptg * ph * pw
                xg = x.view(B, T, hw, self.dim)[:, (row0 // hw):(row0 // hw) + ntg].reshape(rows, self.dim)
                pix = torch.empty((rows, P), device=dev, dtype=torch.float32)
                L.gemm(dt, xg, linear_weight(lin, dt), rows, P, self.dim, C=pix, bias=lin.bias)
                L.unpatchify(pix, video, f0, ntg, ptg, ph, pw)
                continue
            idx, row_off, col_off = maps
            # 'b t h w (c pt p1 p2) -> b c (t pt) (h p1) (w p2)' is separable in (row, column), so the GEMM epilogue writes the video
            # in place (round 1 materialised the (rows, P) pixel matrix -- 100 MB at the bench shape -- and re-read it in pk_unpatchify)
            L.gemm(dt, x, linear_weight(lin, dt), idx.numel(), col_off.numel(), self.dim, C=video, bias=lin.bias, a_rows=idx,
                   scatter=(row_off, col_off))
        return video

    # ---------------------------------------------------------------- public surface (cvivit.py:437-583)

    def decode_from_codebook_indices(self, indices, _prime_indices=None):
        """cvivit.py:437-447.  _prime_indices (B, n_prime) int64: tokens in front of every sequence of `indices` (the sampler's primed
        frames, phenaki_pytorch.py:535-536) without a concatenated copy."""
        L.require_device(indices, 'indices')
        B = indices.shape[0]
        hw = self.image_num_tokens
        flat = indices.reshape(B, -1)
        if flat.dtype != torch.int64:
            flat = flat.long()
        n_tot = flat.shape[1] + (_prime_indices.shape[1] if _prime_indices is not None else 0)
        assert n_tot % hw == 0
        T = n_tot // hw
        # the quantizer writes the code rows directly in the temporal transformer's '(b h w) t' order: no transpose pass
        codes = self.vq.codes_2d(flat, ids_prime=_prime_indices, perm=(T, hw))
        return self._decode2d(codes, B, T, temporal_rows=True)

    def encode(self, tokens):
        L.require_device(tokens, 'tokens')
        B, T, h, w, D = tokens.shape
        out = self._encode2d(tokens.reshape(-1, D).float().contiguous(), B, T)
        return out.view(B, T, h, w, D)

    def decode(self, tokens):
        L.require_device(tokens, 'tokens')
        B = tokens.shape[0]
        h, w = self.patch_height_width
        D = tokens.shape[-1]
        if tokens.ndim == 3:
            T = tokens.shape[1] // (h * w)
        else:
            T = tokens.shape[1]
        return self._decode2d(tokens.reshape(-1, D).float().contiguous(), B, T)

    def tokenize(self, video, return_proj=False):
        """ids (B, T', h, w) int64 [and the pre-sign LFQ projection (B, n, cd), used by the parity margin audit]."""
        tokens, T = self._patch_embed(video)
        tokens_t = self.__dict__.pop('_pk_tokens_t', None)
        B = video.shape[0]
        h, w = self.patch_height_width
        if self.lookup_free_quantization and self.vq.codebook_dim <= 16:
            # the encoder's last LayerNorm (rows '(b h w) t' -> '(b t) (h w)') and the quantizer in one launch
            x = self._encode2d(tokens, B, T, skip_norm_out=True, xt=tokens_t)
            r = self.vq.encode_ids_from_prenorm(x, self.enc_temporal_transformer.norm_out, perm=(h * w, T), return_proj=return_proj)
            if return_proj:
                return r[0].view(B, T, h, w), r[1].view(B, T * h * w, -1)
            return r.view(B, T, h, w)
        tokens = self._encode2d(tokens, B, T, xt=tokens_t)
        if return_proj:
            assert self.lookup_free_quantization, 'the pre-sign projection exists for LFQ only'
            ids, proj = self.vq.encode_ids(tokens, return_proj=True)
            return ids.view(B, T, h, w), proj.view(B, T * h * w, -1)
        return self.vq.encode_ids(tokens).view(B, T, h, w)

    def forward(self, video, mask=None, return_recons=False, return_recons_only=False, return_discr_loss=False,
                apply_grad_penalty=True, return_only_codebook_ids=False):
        if return_discr_loss and not (return_only_codebook_ids or return_recons_only):
            # cvivit.py:604-622: the discriminator's step (hinge + gradient penalty); a graph over self.discr's parameters when grad mode is on
            from .train_cvivit import cvivit_discr_loss
            self._check_video(video, mask)
            return cvivit_discr_loss(self, video, mask=mask, apply_grad_penalty=apply_grad_penalty, return_recons=return_recons)
        if not (return_only_codebook_ids or return_recons_only):
            from .train import wants_grad
            # the GAN objective differentiates its own terms for the adaptive weight (cvivit.py:657-662) and needs the discriminator and the
            # perceptual network: under torch.no_grad(), or on a copy_for_eval() module (both set to None, cvivit.py:415-417), forward(video)
            # is an EVALUATION call and returns the value of the reconstruction loss (the reference raises inside autograd.grad there)
            eval_call = self.use_vgg_and_gan and (not torch.is_grad_enabled() or self.discr is None)
            if eval_call and torch.is_grad_enabled() and not self.__dict__.get('_pk_warned_eval_objective'):
                # a copy_for_eval() module called with grad mode ON returns a DIFFERENT objective than the training module (no
                # perceptual, generator or quantizer terms; the reference raises here) -- say so once instead of silently changing it
                import warnings
                warnings.warn('CViViT.forward(video) on a GAN-mode module without its discriminator / perceptual network (copy_for_eval) returns '
                              'the RECONSTRUCTION loss only, as a value without an autograd graph; the training objective needs the original module',
                              RuntimeWarning, stacklevel=2)
                self.__dict__['_pk_warned_eval_objective'] = True
            if (wants_grad(self) or self.use_vgg_and_gan) and not eval_call:
                # grad mode on and trainable parameters: the tokenizer's training step (train_cvivit.py)
                from .train_cvivit import cvivit_loss_train
                self._check_video(video, mask)
                return cvivit_loss_train(self, video, mask=mask, return_recons=return_recons)
        out = self._forward(video, mask, return_recons, return_recons_only, return_discr_loss, apply_grad_penalty,
                            return_only_codebook_ids)
        if return_only_codebook_ids or return_recons_only:
            return out
        return value_without_graph(self, 'CViViT.forward (reconstruction loss)', out)

    def _check_video(self, video, mask):
        """the asserts of cvivit.py:529-540"""
        assert video.ndim in {4, 5}
        if video.ndim == 4:
            video = video.unsqueeze(2)
            assert not exists(mask)
        b, c, f, *image_dims = video.shape
        assert tuple(image_dims) == self.image_size
        assert not exists(mask) or mask.shape[-1] == f
        assert divisible_by(f - 1, self.temporal_patch_size), \
            f'number of frames ({f}) minus one ({f - 1}) must be divisible by temporal patch size ({self.temporal_patch_size})'
        return video

    @torch.no_grad()
    def _forward(self, video, mask=None, return_recons=False, return_recons_only=False, return_discr_loss=False,
                 apply_grad_penalty=True, return_only_codebook_ids=False):
        is_image = video.ndim == 4
        video = self._check_video(video, mask)
        b, c, f, *image_dims = video.shape
        L.require_device(video, 'video')
        video = video.float().contiguous()

        if return_only_codebook_ids:
            return self.tokenize(video)

        assert return_recons_only or not return_discr_loss, 'forward() routes the discriminator objective to train_cvivit.py'
        ids = self.tokenize(video).reshape(-1)
        T = 1 + (f - 1) // self.temporal_patch_size
        recon = self._decode2d(self.vq.codes_2d(ids, perm=(T, self.image_num_tokens)), b, T, temporal_rows=True)
        returned_recon = recon.squeeze(2) if is_image else recon
        if return_recons_only:
            return returned_recon
        # cvivit.py:585-627 with use_vgg_and_gan = False: the VALUE of the reconstruction loss (no autograd graph),
        # F.mse_loss over all pixels or over the frames `mask` (b, f) keeps
        if exists(mask):
            L.require_device(mask, 'mask')
            count = mask.sum().double() * (c * image_dims[0] * image_dims[1])
        else:
            count = float(video.numel())
        recon_loss = (L.sqdiff_sum(video, recon.contiguous(), mask) / count).float()
        if return_recons:
            return recon_loss, returned_recon
        return recon_loss
