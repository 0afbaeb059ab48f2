"""The training step of the reference's CViViTTrainer (cvivit_trainer.py:226-270) on the MI355X kernels.  Default: the generator step without
the GAN terms -- zero_grad -> CViViT.forward (reconstruction MSE through the straight-through LFQ) -> backward -> [gradient all-reduce] -> AdamW.
`--gan`: both halves of train_step -- the generator objective recon + perceptual + adaptive_weight * hinge generator loss (cvivit.py:585-671; the
perceptual network is whatever module is passed as `vgg=`, here a small stand-in because torchvision's VGG16 cannot be downloaded offline), then the
discriminator's hinge loss with the gradient penalty every `--gp-every`-th step (cvivit_trainer.py:224, 251-270).  Finally the
reconstruction of one batch is written as a GIF (data.py:103-113).  `--folder` trains on the GIFs of a directory through VideoDataset /
DataLoader (data.py:177-265); without it synthetic videos stand in.  One process per GPU (`torchrun --nproc-per-node N ...`).

    python examples/train_cvivit.py --steps 20 --small --gif /tmp/recon.gif
    python examples/train_cvivit.py --steps 20 --small --gan
"""
import argparse
import contextlib
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phenaki_pytorch_amd as P  # noqa: E402
from phenaki_pytorch_amd.data import DataLoader, VideoDataset, video_tensor_to_gif  # noqa: E402


def perceptual_stand_in(size):
    """any nn.Module mapping (B, 3, H, W) frames to features serves as `vgg=` (cvivit.py:346-347); offline there is no pretrained VGG16"""
    from torch import nn
    net = nn.Sequential(nn.AvgPool2d(4), nn.Flatten(), nn.Linear(3 * (size // 4) ** 2, 256), nn.Tanh(), nn.Linear(256, 64))
    for p in net.parameters():
        p.requires_grad_(False)
    return net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--frames', type=int, default=17)
    ap.add_argument('--dtype', default='bf16x3', choices=['fp32', 'bf16x3', 'bf16'])
    ap.add_argument('--small', action='store_true', help='a small geometry (dim 128, 64 x 64 pixels) instead of the BASELINE one')
    ap.add_argument('--folder', default='', help='directory of GIFs to train on (VideoDataset); default: synthetic videos')
    ap.add_argument('--gif', default='', help='write the reconstruction of the last batch here')
    ap.add_argument('--save', default='')
    ap.add_argument('--gan', action='store_true', help='train with the perceptual + adversarial objective and a discriminator step')
    ap.add_argument('--gp-every', type=int, default=4, help='apply the gradient penalty every this many steps (cvivit_trainer.py:224)')
    args = ap.parse_args()
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    rank = dist.get_rank() if ws > 1 else 0
    torch.manual_seed(1)                   # the model is built under a rank-INDEPENDENT seed (and GradientReducer broadcasts rank 0's weights anyway)

    dim, size, patch, vocab = (128, 64, 16, 256) if args.small else (512, 256, 32, 65536)
    cvivit = P.CViViT(dim=dim, codebook_size=vocab, image_size=size, patch_size=patch, temporal_patch_size=2, spatial_depth=2 if args.small else 4,
                      temporal_depth=2 if args.small else 4, dim_head=64, heads=dim // 64, use_vgg_and_gan=args.gan,
                      vgg=perceptual_stand_in(size) if args.gan else None).cuda().train()
    P.set_compute_dtype(cvivit, args.dtype)
    # the reference keeps two optimizers: the tokenizer's parameters (everything but discr.*) and the discriminator's (cvivit_trainer.py:118-124)
    params = [p for n, p in cvivit.named_parameters() if p.requires_grad and not n.startswith('discr.')]
    opt = P.get_optimizer(params, lr=3e-4, wd=0.)
    discr_params = list(cvivit.discr.parameters()) if args.gan else []
    discr_opt = P.get_optimizer(discr_params, lr=3e-4, wd=0.) if args.gan else None
    reducer = P.GradientReducer(params, buffers=list(cvivit.buffers())) if ws > 1 else None     # broadcasts rank 0's parameters, as DDP does at wrap time
    discr_reducer = P.GradientReducer(discr_params) if (ws > 1 and args.gan) else None
    torch.manual_seed(1 + rank)            # from here on (data order, frame masks) every rank draws its own stream

    if args.folder:
        loader = DataLoader(VideoDataset(args.folder, size, num_frames=args.frames), batch_size=args.batch, shuffle=True, drop_last=True)

        def batches():
            while True:
                for (clip,) in loader:
                    yield clip.cuda()
        stream = batches()
    else:
        fixed = torch.rand(args.batch, 3, args.frames, size, size, device='cuda')
        stream = iter(lambda: fixed, None)

    t0 = None
    for step in range(args.steps):
        if step == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        videos = next(stream)
        opt.zero_grad(set_to_none=True)
        loss = cvivit(videos)
        # the generator objective also reaches discr.* through gen_loss: those gradients are dropped below (discr_opt.zero_grad), so their
        # reducer must not start collectives for them -- a launched bucket would meet the discriminator step's own backward
        with (discr_reducer.no_sync() if discr_reducer is not None else contextlib.nullcontext()):
            loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        discr_loss = None
        if args.gan:
            discr_opt.zero_grad(set_to_none=True)                      # also drops what the generator objective left in discr.*.grad
            discr_loss = cvivit(next(stream), return_discr_loss=True, apply_grad_penalty=(step % args.gp_every == 0))
            discr_loss.backward()
            if discr_reducer is not None:
                discr_reducer.finish()
            discr_opt.step()
        if rank == 0 and (step % 5 == 0 or step == args.steps - 1):
            tail = f'  discriminator loss {float(discr_loss.detach()):.5f}' if discr_loss is not None else ''
            print(f'step {step:4d}  {"vae" if args.gan else "reconstruction"} loss {float(loss.detach()):.5f}{tail}', flush=True)
    torch.cuda.synchronize()
    if rank == 0 and t0 is not None and args.steps > 3:
        dt = (time.perf_counter() - t0) / (args.steps - 3)
        print(f'{dt * 1e3:.1f} ms per step, {args.batch * args.frames * ws / dt:.0f} frames/s')
    if rank == 0 and args.gif:
        with torch.no_grad():
            recon = cvivit.eval()(videos[:1], return_recons_only=True)
        video_tensor_to_gif(recon[0].clamp(0, 1).cpu(), args.gif)
        print('wrote', args.gif)
    if rank == 0 and args.save:
        torch.save(cvivit.state_dict(), args.save)
    if ws > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
