"""diagnostic: where does the discriminator's gradient error against the CPU oracle come from?  (prints relative L2 errors per parameter)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gan_oracle as G, weights
import phenaki_pytorch_amd as P
from phenaki_pytorch_amd.discriminator import Discriminator, gradient_penalty

def rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().reshape(-1)
    e = (a - b).abs()
    return float(e.norm() / (b.norm() + 1e-30)), float(e.max() / (b.abs().max() + 1e-30)), float((e > 1e-3 * b.abs().max()).double().mean())

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
d = Discriminator(dim=16, image_size=size)
weights.fill_module(d, salt=1)
x = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(10))
for what in ('sum', 'gp'):
    for so in ((False, True) if what == 'sum' else (True,)):
        sd = {'discr.' + k: v.detach().clone().requires_grad_(v.is_floating_point() and not k.endswith('beta')) for k, v in d.state_dict().items()}
        xo = x.clone().requires_grad_()
        lo = G.discriminator(sd, xo)
        obj = lo.sum() if what == 'sum' else G.gradient_penalty(xo, lo)
        gx_o, = torch.autograd.grad(obj, xo, retain_graph=True)
        obj.backward()
        dd = Discriminator(dim=16, image_size=size)
        dd.load_state_dict(d.state_dict())
        dd = dd.cuda()
        P.set_compute_dtype(dd, mode)
        xp = x.cuda().requires_grad_()
        lp = dd(xp, second_order=so)
        objp = lp.sum() if what == 'sum' else gradient_penalty(xp, lp)
        gx_p, = torch.autograd.grad(objp, xp, retain_graph=True)
        objp.backward()
        print(f'== {what} second_order={so} mode={mode}: logits {rel(lp, lo)}  objective {float(objp.detach()):.7f} vs {float(obj.detach()):.7f}  d/dx {rel(gx_p, gx_o)}')
        for k, v in dd.named_parameters():
            r = sd['discr.' + k].grad
            if r is None or not r.numel():
                continue
            print(f'   {k:40s} relL2 {rel(v.grad, r)[0]:.2e} max {rel(v.grad, r)[1]:.2e} frac {rel(v.grad, r)[2]:.2e}')
