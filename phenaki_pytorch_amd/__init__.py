"""phenaki_pytorch_amd -- the Phenaki hot path (C-ViViT tokenizer + MaskGIT sampler; the MaskGit / critic training step) on MI355X (gfx950).

Drop-in names of lucidrains/phenaki-pytorch's public interface (phenaki_pytorch/__init__.py:1-4): same constructor signatures,
state_dict keys and .forward/.encode/.decode/.sample surfaces; `Phenaki.forward` under grad mode is the training step (train.py,
optim.py); the compute is hand-written HIP in libphenaki_hip.so (build: `python -m phenaki_pytorch_amd.build`).
"""
from .attention import set_compute_dtype, invalidate_packed
from .cvivit import CViViT
from .phenaki import MaskGit, TokenCritic, SelfCritic, Phenaki, make_video
from .dist import shard_batch, sample_sharded, make_video_sharded, all_reduce_gradients, GradientReducer, broadcast_parameters, broadcast_module
from .train import vocab_cross_entropy
from .optim import HipAdamW, get_optimizer

__all__ = ['CViViT', 'MaskGit', 'TokenCritic', 'SelfCritic', 'Phenaki', 'make_video', 'set_compute_dtype', 'invalidate_packed',
           'shard_batch', 'sample_sharded', 'make_video_sharded', 'vocab_cross_entropy', 'all_reduce_gradients', 'GradientReducer', 'broadcast_parameters', 'broadcast_module', 'HipAdamW', 'get_optimizer']
