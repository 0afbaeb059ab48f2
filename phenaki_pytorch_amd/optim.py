"""Optimizer of the training step on the MI355X kernels -- reference /root/reference/phenaki_pytorch/optimizer.py:1-37 (`get_optimizer`:
Adam when wd == 0, else AdamW with the parameters of fewer than 2 dimensions excluded from the decay), called by phenaki_trainer.py:284.
`HipAdamW` is a torch.optim.Optimizer (state_dict / param_groups / zero_grad as usual) whose update is pk_adamw_multi: the large tensors one launch each, the hundreds of small ones packed 40 per launch."""
import torch

from . import _lib as L


def separate_weight_decayable_params(params):
    wd_params, no_wd_params = [], []
    for param in params:
        (no_wd_params if param.ndim < 2 else wd_params).append(param)
    return wd_params, no_wd_params


class HipAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled decay, bias-corrected moments, eps outside the root); weight_decay = 0 is torch.optim.Adam"""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group['betas']
            batches = {}                                        # step number -> [(p, g, m, v)]: one pk_adamw_multi call per (group, step)
            for p in group['params']:
                if p.grad is None or p.numel() == 0:            # (the self-attention blocks carry an empty null_kv)
                    continue
                L.require_device(p, 'parameter')
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError('HipAdamW updates contiguous float32 parameters (keep the modules in float32)')
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                g = g.float() if g.dtype != torch.float32 else g
                batches.setdefault((st['step'], p.device), []).append((p, g, st['exp_avg'], st['exp_avg_sq']))
            for (step, device), entries in batches.items():
                # the small tensors of the batch (LayerNorm gains, biases, scales: most of a transformer's parameter LIST) share launches
                L.adamw_multi(entries, group['lr'], b1, b2, group['eps'], group['weight_decay'], step, device)
                for p, _, _, _ in entries:
                    torch.autograd.graph.increment_version(p)  # the kernel wrote through the raw pointer: packed-weight caches / captured graphs key on _version
        return loss


def get_optimizer(params, lr=1e-4, wd=1e-2, betas=(0.9, 0.99), eps=1e-8, filter_by_requires_grad=False, group_wd_params=True, **kwargs):
    params = list(params)
    if filter_by_requires_grad:
        params = [t for t in params if t.requires_grad]
    if wd == 0:
        return HipAdamW(params, lr=lr, betas=betas, eps=eps, weight_decay=0.)
    if group_wd_params:
        wd_params, no_wd_params = separate_weight_decayable_params(params)
        params = [{'params': wd_params}, {'params': no_wd_params, 'weight_decay': 0}]
    return HipAdamW(params, lr=lr, weight_decay=wd, betas=betas, eps=eps)
