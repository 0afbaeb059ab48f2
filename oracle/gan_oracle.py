"""CPU restatement of the tokenizer's adversarial branch (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py): the Discriminator of
/root/reference/phenaki_pytorch/cvivit.py:101-213, the losses of :59-99 and the two objectives CViViT.forward builds from them (:593-671),
as plain torch functions over a name-keyed state_dict (`discr.*` beside the C-ViViT entries).  Pinned against the real reference by
tests/golden/gan_tiny.pt (oracle/make_golden.py gan_golden; tests/test_oracle_golden.py)."""
import math

import torch
import torch.nn.functional as F

from oracle import phenaki_oracle as O


def pick_video_frame(video, frame):
    """cvivit.py:217-224: (B, C, F, H, W), frame (B,) -> (B, C, H, W)"""
    return video[torch.arange(video.shape[0]), :, frame.long()]


def discriminator(sd, x, p='discr.', heads=8):
    """cvivit.py:200-213 (+ DiscriminatorBlock.forward :129-138): (B, C, H, W) -> (B,) logits"""
    i = 0
    while f'{p}blocks.{i}.conv_res.weight' in sd:
        bp = f'{p}blocks.{i}.'
        down = f'{bp}downsample.1.weight' in sd
        res = F.conv2d(x, sd[bp + 'conv_res.weight'], sd[bp + 'conv_res.bias'], stride=2 if down else 1)
        h = F.leaky_relu(F.conv2d(x, sd[bp + 'net.0.weight'], sd[bp + 'net.0.bias'], padding=1), 0.1)
        h = F.leaky_relu(F.conv2d(h, sd[bp + 'net.2.weight'], sd[bp + 'net.2.bias'], padding=1), 0.1)
        if down:
            b, c, H, W = h.shape                                       # Rearrange('b c (h p1) (w p2) -> b (c p1 p2) h w', p1 = 2, p2 = 2)
            h = h.reshape(b, c, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, H // 2, W // 2)
            h = F.conv2d(h, sd[bp + 'downsample.1.weight'], sd[bp + 'downsample.1.bias'])
        x = (h + res) * (1 / math.sqrt(2))
        ap = f'{p}attn_blocks.{i}.'
        if ap + 'to_q.weight' in sd:
            b, c, H, W = x.shape
            t = x.flatten(2).transpose(1, 2)                            # 'b c n -> b n c'
            t = O.attention(sd, ap, t, heads=heads) + t
            x = t.transpose(1, 2).reshape(b, c, H, W)
        i += 1
    x = F.leaky_relu(F.conv2d(x, sd[p + 'to_logits.0.weight'], sd[p + 'to_logits.0.bias'], padding=1), 0.1)
    return (x.flatten(1) @ sd[p + 'to_logits.3.weight'].t() + sd[p + 'to_logits.3.bias']).squeeze(-1)


def gradient_penalty(images, output, weight=10):
    """cvivit.py:59-73"""
    g, = torch.autograd.grad(outputs=output, inputs=images, grad_outputs=torch.ones_like(output), create_graph=True, retain_graph=True)
    return weight * ((g.reshape(images.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean()


def hinge_discr_loss(fake, real):
    return (F.relu(1 + fake) + F.relu(1 - real)).mean()


def hinge_gen_loss(fake):
    return -fake.mean()


def discr_loss(sd, cfg, video, frame, apply_grad_penalty=True):
    """CViViT.forward(video, return_discr_loss=True), cvivit.py:604-622, for the frame choice `frame` (B,)"""
    with torch.no_grad():
        recon = O.cvivit_reconstruct_train(sd, cfg, video)
    real = pick_video_frame(video, frame).detach().requires_grad_()
    fake = pick_video_frame(recon, frame).detach()
    fake_logits, real_logits = discriminator(sd, fake), discriminator(sd, real)
    loss = hinge_discr_loss(fake_logits, real_logits)
    if apply_grad_penalty:
        loss = loss + gradient_penalty(real, real_logits)
    return loss


def generator_loss(sd, cfg, video, frame, vgg, mask=None, parts=None):
    """CViViT.forward(video) with use_vgg_and_gan=True, cvivit.py:585-671: recon + perceptual + vq_aux + adaptive_weight * gen
    (vq_aux: the LFQ's training-mode entropy + commitment loss with the published defaults, oracle/lfq.py lfq_aux_loss; cfg['lfq_kwargs'] overrides)"""
    from oracle import lfq
    recon, proj = O.cvivit_reconstruct_train(sd, cfg, video, return_proj=True)
    bd = {}
    vq_aux = lfq.lfq_aux_loss(proj, breakdown=bd, **cfg.get('lfq_kwargs', {}))
    recon_loss = O._masked_mse(video, recon, mask)
    real_img, recon_img = pick_video_frame(video, frame), pick_video_frame(recon, frame)
    perceptual = F.mse_loss(vgg(real_img), vgg(recon_img))
    gen = hinge_gen_loss(discriminator(sd, recon_img))
    last = sd['to_pixels.0.weight']
    n_gen = torch.autograd.grad(gen, last, retain_graph=True)[0].detach().norm(p=2)
    n_per = torch.autograd.grad(perceptual, last, retain_graph=True)[0].detach().norm(p=2)
    adaptive = (n_per / (n_gen + 1e-8)).clamp(max=1e4)
    if parts is not None:
        parts.update(recon_loss=recon_loss.detach(), perceptual=perceptual.detach(), gen_loss=gen.detach(), adaptive_weight=adaptive.detach(),
                     vq_aux=vq_aux.detach(), **bd)
    return recon_loss + perceptual + vq_aux + adaptive * gen
