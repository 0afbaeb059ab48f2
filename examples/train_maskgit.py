"""The inner loop of the reference's PhenakiTrainer (phenaki_trainer.py:351-388) on the MI355X kernels: zero_grad -> Phenaki.forward (MaskGit
cross entropy + TokenCritic BCE) -> backward -> [gradient all-reduce] -> AdamW.  Synthetic videos and random text embeddings stand in for a
dataset and the T5 encoder; one process per GPU (`torchrun --nproc-per-node N examples/train_maskgit.py` averages gradients over RCCL).

    python examples/train_maskgit.py --steps 20 --dtype bf16x3
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phenaki_pytorch_amd as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--dtype', default='bf16x3', choices=['fp32', 'bf16x3', 'bf16'])
    ap.add_argument('--small', action='store_true', help='a small geometry (dim 128, 64 x 64 pixels) instead of the BASELINE one')
    ap.add_argument('--save', default='')
    args = ap.parse_args()
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    rank = dist.get_rank() if ws > 1 else 0
    torch.manual_seed(1)                   # models (incl. the frozen tokenizer) are built under a rank-INDEPENDENT seed: every rank tokenizes alike

    dim, size, patch, vocab, ctx_dim = (128, 64, 16, 256, 96) if args.small else (512, 256, 32, 65536, 768)
    cvivit = P.CViViT(dim=dim, codebook_size=vocab, image_size=size, patch_size=patch, temporal_patch_size=2, spatial_depth=2 if args.small else 4,
                      temporal_depth=2 if args.small else 4, dim_head=64, heads=dim // 64, use_vgg_and_gan=False)
    maskgit = P.MaskGit(dim=dim, num_tokens=vocab, max_seq_len=1024, depth=2 if args.small else 6, heads=dim // 64, dim_head=64, dim_context=ctx_dim)
    critic = P.TokenCritic(dim=dim, num_tokens=vocab, max_seq_len=1024, depth=2 if args.small else 6, heads=dim // 64, dim_head=64, dim_context=ctx_dim,
                           has_cross_attn=True)
    phenaki = P.Phenaki(cvivit=cvivit, maskgit=maskgit, critic=critic, text_embed_dim=ctx_dim).cuda()
    P.set_compute_dtype(phenaki, args.dtype)
    params = list(maskgit.parameters()) + list(critic.parameters())
    opt = P.get_optimizer(params, lr=1e-4, wd=1e-2)

    if ws > 1:
        P.broadcast_module(phenaki)        # every replica starts from rank 0's weights and buffers (what accelerate / DDP do at wrap time)
    torch.manual_seed(1 + rank)            # from here on (data, mask draws) every rank has its own stream
    videos = torch.randn(args.batch, 3, 17, size, size, device='cuda')          # a dataset would go here
    text_embeds = torch.randn(args.batch, 12, ctx_dim, device='cuda')            # ... and phenaki.encode_texts(texts) (t5.py / T5Encoder)
    with torch.no_grad():
        ids = phenaki.cvivit(videos, return_only_codebook_ids=True)              # the tokenizer is frozen: encode once per batch
    reducer = P.GradientReducer(params, broadcast=False) if ws > 1 else None
    t0 = None
    for step in range(args.steps):
        if step == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = phenaki(video_codebook_ids=ids, text_embeds=text_embeds)
        loss.backward()
        if reducer is not None:
            reducer.finish()                                                     # the buckets were all-reduced while backward ran (dist.py)
        opt.step()
        if step % 5 == 0 or step == args.steps - 1:
            print(f'step {step:4d}  loss {float(loss.detach()):.4f}', flush=True)
    torch.cuda.synchronize()
    if t0 is not None and args.steps > 3:
        dt = (time.perf_counter() - t0) / (args.steps - 3)
        print(f'{dt * 1e3:.1f} ms per step, {args.batch * ws / dt:.1f} videos/s')
    if args.save and (ws == 1 or dist.get_rank() == 0):
        torch.save(dict(maskgit=maskgit.state_dict(), critic=critic.state_dict(), optimizer=opt.state_dict()), args.save)      # reference-compatible keys
    if ws > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
