"""The generator step of the reference's CViViTTrainer (cvivit_trainer.py:241-249) without the GAN terms, on the MI355X kernels:
zero_grad -> CViViT.forward (reconstruction MSE through the straight-through LFQ) -> backward -> [gradient all-reduce] -> AdamW, then the
reconstruction of one batch written as a GIF (data.py:103-113).  `--folder` trains on the GIFs of a directory through VideoDataset /
DataLoader (data.py:177-265); without it synthetic videos stand in.  One process per GPU (`torchrun --nproc-per-node N ...`).

    python examples/train_cvivit.py --steps 20 --small --gif /tmp/recon.gif
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phenaki_pytorch_amd as P  # noqa: E402
from phenaki_pytorch_amd.data import DataLoader, VideoDataset, video_tensor_to_gif  # noqa: E402


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
    args = ap.parse_args()
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    rank = dist.get_rank() if ws > 1 else 0
    torch.manual_seed(1)                   # the model is built under a rank-INDEPENDENT seed (and GradientReducer broadcasts rank 0's weights anyway)

    dim, size, patch, vocab = (128, 64, 16, 256) if args.small else (512, 256, 32, 65536)
    cvivit = P.CViViT(dim=dim, codebook_size=vocab, image_size=size, patch_size=patch, temporal_patch_size=2, spatial_depth=2 if args.small else 4,
                      temporal_depth=2 if args.small else 4, dim_head=64, heads=dim // 64, use_vgg_and_gan=False).cuda().train()
    P.set_compute_dtype(cvivit, args.dtype)
    params = list(cvivit.parameters())
    opt = P.get_optimizer(params, lr=3e-4, wd=0.)
    reducer = P.GradientReducer(params, buffers=list(cvivit.buffers())) if ws > 1 else None     # broadcasts rank 0's parameters, as DDP does at wrap time
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
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        if rank == 0 and (step % 5 == 0 or step == args.steps - 1):
            print(f'step {step:4d}  reconstruction loss {float(loss.detach()):.5f}', flush=True)
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
