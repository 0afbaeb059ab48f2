"""bench.py -- the BASELINE.json metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE.json configs[1], the C-ViViT
tokenizer (dim 512, patch 32, temporal patch 2, depth 4+4, LFQ 65 536) encoding a (8, 3, 17, 256, 256) f32 video that
is already resident in HBM into (8, 9, 8, 8) int64 token ids, bf16 MFMA operands / f32 accumulation.  With N > 1 every
rank encodes its own 8 videos (batch sharding, weak scaling, no collective on the data path); the timed region is
bracketed by barrier + synchronize and the reported time is the max over ranks.

The same JSON line also carries
  roofline     : the dominant kernel (the MFMA GEMM) -- algorithmic flops / launch over the live HIP-event duration;
  cpu_baseline : the CPU oracle (a port of the reference algorithm, oracle/phenaki_oracle.py) on a bounded sample of
                 the same workload on this box's host cores;
  sample       : the second half of the metric, MaskGIT sampled tokens/sec of an 18-step Phenaki.sample (configs[2]).
"""
import argparse
import json
import os
import sys
import time

# kernel arguments in device memory: the ROCm 7.2 default on this GPU, pinned here because the sampling loop is ~3 000 short
# launches per call (measured: 65.6 ms with it, 70.6 ms with HIP_FORCE_DEV_KERNARG=0); must be set before HIP initialises
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8, help='videos per GPU (configs[1]: 8)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--no-sample', action='store_true', help='skip the MaskGIT sampling leg')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--sample-batch', type=int, default=8)
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    return ap.parse_args()


def init_dist(n):
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    ws = int(os.environ.get('WORLD_SIZE', 1))
    if ws > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', rank=rank, world_size=ws, device_id=torch.device('cuda', local))
    else:
        torch.cuda.set_device(0)
    assert ws == n or ws == 1, f'--gpus {n} but WORLD_SIZE={ws}'
    return rank, local, ws


def barrier_sync(ws):
    if ws > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, ws):
    if ws == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], device='cuda', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# BASELINE.json configs[1..2]: C-ViViT dim 512 / codebook 65 536 / 256x256 / patch 32 / temporal patch 2 / depth 4+4;
# MaskGit and TokenCritic dim 512 / depth 6 / 8 heads / T5-base context (768); 18 sampling steps
BASELINE_CFG = dict(
    cvivit=dict(dim=512, codebook_size=65536, image_size=256, patch_size=32, temporal_patch_size=2,
                spatial_depth=4, temporal_depth=4, dim_head=64, heads=8),
    maskgit=dict(dim=512, num_tokens=65536, max_seq_len=1024, depth=6, heads=8, dim_head=64, dim_context=768),
    critic=dict(dim=512, num_tokens=65536, max_seq_len=1024, depth=6, heads=8, dim_head=64, dim_context=768,
                has_cross_attn=True),
    steps=18,
)


def build_models(dtype, with_sampler):
    """random-init weights of the BASELINE architecture (the modules' own default initialisers under a fixed seed; no
    checkpoints offline).  The GPU legs of this file use the product package only -- nothing under oracle/ or tests/."""
    import phenaki_pytorch_amd as P
    torch.manual_seed(0)
    cv = P.CViViT(use_vgg_and_gan=False, **BASELINE_CFG['cvivit'])
    mg = P.MaskGit(**BASELINE_CFG['maskgit']) if with_sampler else None
    cr = P.TokenCritic(**BASELINE_CFG['critic']) if with_sampler else None
    cv = cv.cuda().eval()
    ph = None
    if with_sampler:
        mg, cr = mg.cuda().eval(), cr.cuda().eval()
        ph = P.Phenaki(maskgit=mg, cvivit=cv, critic=cr, steps=BASELINE_CFG['steps'],
                       text_embed_dim=BASELINE_CFG['maskgit']['dim_context']).cuda().eval()
    for m in (cv, mg, cr, ph):
        if m is not None:
            P.set_compute_dtype(m, dtype)
    return cv, mg, cr, ph


def synthetic_video(batch, frames, size, seed):
    g = torch.Generator(device='cpu')
    g.manual_seed(1000 + seed)
    return torch.randn(batch, 3, frames, size, size, generator=g)


def synthetic_context(batch, length, dim, seed):
    """stands in for the cached T5 encoder output (no all-zero rows: every context token is real)"""
    g = torch.Generator(device='cpu')
    g.manual_seed(2000 + seed)
    return torch.randn(batch, length, dim, generator=g)


# template order: T, TM, TN, WM, WN, STAGES, ROWB, PW (the names rocprofv3 prints)
VARIANT_KERNEL = {1: 'gemm_kernel<T,TA,2,2>', 2: 'gemm_kernel<T,TA,4,4>', 3: 'gemm_dma_kernel<T,2,2,2,2,4,128,0>',
                  8: 'gemm_dma_kernel<T,2,2,2,2,2,128,0>', 9: 'gemm_dma_kernel<T,4,4,2,2,2,128,0>',
                  24: 'gemm_dma_kernel<T,4,2,2,4,2,128,0>', 33: 'gemm_dma_kernel<T,2,2,2,2,3,128,2>'}


class GemmProfiler:
    """live HIP-event timing of the pk_gemm launches of one (untimed, eager) pass of the step, on the stream the
    kernels run on: every launch of the pass is recorded with its live operands, then each one is re-issued REPS times
    back to back between two HIP events (a per-launch event pair around a ~10 us kernel mostly measures the events).
    Grouped by the kernel instantiation pk_gemm picked (the names rocprofv3 reports)."""
    REPS = 20

    def __init__(self):
        self.calls = []

    def __enter__(self):
        from phenaki_pytorch_amd import _lib
        self._lib = _lib
        self._orig = _lib.gemm
        prof = self

        def gemm(dtype, A, W, M, N, K, **kw):
            prof.calls.append((dtype, A, W, M, N, K, kw))
            return prof._orig(dtype, A, W, M, N, K, **kw)
        _lib.gemm = gemm
        return self

    def __exit__(self, *exc):
        self._lib.gemm = self._orig
        torch.cuda.synchronize()

    def summary(self):
        lib, by = self._lib, {}
        for dtype, A, W, M, N, K, kw in self.calls:
            a_is_f32 = 1 if A.dtype == torch.float32 else 0
            rows = A.shape[0] if kw.get('a_rows') is not None else M
            v = lib.load().pk_gemm_auto_variant(dtype, a_is_f32, M, N, K, A.stride(-2), W.stride(0), rows)
            t = 'pk::bf16' if dtype == lib.BF16 else 'float'
            name = VARIANT_KERNEL.get(v, f'variant{v}').replace('TA', 'float' if a_is_f32 else t).replace('T', t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._orig(dtype, A, W, M, N, K, **kw)
            e0.record()
            for _ in range(self.REPS):
                self._orig(dtype, A, W, M, N, K, **kw)
            e1.record()
            torch.cuda.synchronize()
            d = by.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += 2.0 * M * N * K
            d[2] += e0.elapsed_time(e1) * 1e-3 / self.REPS
        return {k: dict(launches=v[0], flops=v[1], seconds=v[2]) for k, v in by.items()}


def pmc_traffic(kernel, args):
    """HBM bytes per launch of the encode leg's dominant kernel.  PMC counters need their own rocprofv3 passes (they cannot
    be collected from inside this process), so this is the committed measurement of exactly this configuration
    (profiles/gemm_hbm_pmc_r01.txt: mean of the 8 to_out + 8 FF2 launches of a step), or None for any other."""
    if kernel == 'gemm_dma_kernel<pk::bf16,2,2,2,2,2,128,0>' and args.dtype == 'bf16' and args.batch == 8:
        return 35.57e6
    return None


def bench_encode(cv, args, ws):
    B = args.batch
    video = synthetic_video(B, 17, 256, seed=int(os.environ.get('RANK', 0))).cuda()
    # the metric's own entry point: CViViT.forward(video, return_only_codebook_ids=True)
    step = lambda: cv(video, return_only_codebook_ids=True)
    ids = step()                                   # builds the packed weights / bias caches
    torch.cuda.synchronize()
    used_graph = False
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            with torch.cuda.graph(graph):
                ids_g = step()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(ids_g, ids), 'graph replay changed the token ids'
            step = graph.replay
            used_graph = True
        except Exception as e:                     # noqa: BLE001  (recorded in the JSON, never silent)
            print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly', file=sys.stderr)
            torch.cuda.synchronize()
            step = lambda: cv(video, return_only_codebook_ids=True)
    for _ in range(args.warmup):
        step()
    barrier_sync(ws)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier_sync(ws)
    dt = max_over_ranks(time.perf_counter() - t0, ws)
    # roofline of the dominant kernel: one extra, untimed, eager pass with HIP events around every GEMM launch
    with GemmProfiler() as prof:
        cv(video, return_only_codebook_ids=True)
    return dt, used_graph, prof.summary(), ids


def bench_sample(ph, args, ws):
    """configs[2]: 18-step MaskGIT sampling (CFG scale 5, TokenCritic) with frozen random C-ViViT codes and a cached
    (random) T5 context; tokens/sec = B * 576 / wall time of Phenaki.sample (including the final decode)."""
    B = args.sample_batch
    ctx = synthetic_context(B, 12, 768, seed=1).cuda()
    ph.encode_texts = lambda texts, output_device=None: ctx
    texts = ['x'] * B
    torch.manual_seed(0)
    ph.sample(texts=texts, num_frames=17, cond_scale=5.)          # warm-up (packs weights, position bias)
    barrier_sync(ws)
    runs = 2
    t0 = time.perf_counter()
    for _ in range(runs):
        ph.sample(texts=texts, num_frames=17, cond_scale=5.)
    barrier_sync(ws)
    dt = max_over_ranks(time.perf_counter() - t0, ws) / runs
    out = dict(metric='maskgit_sampled_tokens_per_sec', value=B * 576 * ws / dt, unit='tokens/s', seconds_per_sample_call=dt,
               batch_per_gpu=B, steps=ph.steps, cond_scale=5.0, critic='TokenCritic depth 6 cross-attn', tokens_per_video=576,
               noise='in-kernel counter hash (FAST mode)')
    # roofline of this leg's dominant kernel (the MFMA GEMM at M = 2B*576 rows): 2 untimed sampling steps under the profiler
    steps = ph.steps
    try:
        ph.steps = 2
        with GemmProfiler() as prof:
            ph.sample(texts=texts, num_frames=17, cond_scale=5.)
        g = prof.summary()
    finally:
        ph.steps = steps
    if g:
        name, d = max(g.items(), key=lambda kv: kv[1]['seconds'])
        ach = d['flops'] / d['seconds'] / 1e12
        peak = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else 157.3
        out['roofline'] = {'bound': 'mfma', 'kernel': name, 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                           'traffic': None, 'avg_launch_us': d['seconds'] / d['launches'] * 1e6,
                           'algorithmic_flops_per_launch': d['flops'] / d['launches'],
                           'all_gemm_variants': {k: {'launches': v['launches'], 'TFLOP/s': v['flops'] / v['seconds'] / 1e12,
                                                     'us_total': v['seconds'] * 1e6} for k, v in g.items()}}
    return out


def bench_objective(ph, args, ws):
    """SURVEY.md 8f row 1, first slice: Phenaki.forward -- the VALUE of the training objective (masked cross entropy without
    logits + token-critic BCE) on B videos' worth of token ids; videos/sec.  Reported beside the two headline legs."""
    B = args.sample_batch
    ctx = synthetic_context(B, 12, 768, seed=1).cuda()
    g = torch.Generator(device='cpu')
    g.manual_seed(4)
    ids = torch.randint(0, 65536, (B, 9, 8, 8), generator=g).cuda()
    torch.manual_seed(0)
    ph(video_codebook_ids=ids, text_embeds=ctx)                 # warm-up
    barrier_sync(ws)
    runs = 5
    t0 = time.perf_counter()
    for _ in range(runs):
        loss = ph(video_codebook_ids=ids, text_embeds=ctx)
    barrier_sync(ws)
    dt = max_over_ranks(time.perf_counter() - t0, ws) / runs
    return dict(metric='phenaki_forward_objective_videos_per_sec', value=B * ws / dt, unit='videos/s', ms_per_call=dt * 1e3,
                batch_per_gpu=B, tokens_per_video=576, loss=float(loss),
                note='forward value only (no autograd graph): MaskGit trunk + vocab head with fused gumbel sampling and '
                     'cross entropy (logits never written) + TokenCritic trunk + BCE; random-init weights, random ids')


def cpu_baseline(args):
    """the CPU oracle (port of the reference algorithm) on a bounded sample: B = 2 videos per call, repeated for
    ~cpu_seconds; frames/sec on this box's host cores."""
    from oracle import phenaki_oracle as O
    from oracle import weights
    from oracle.configs import FULL, oracle_cfgs, state_dicts
    from oracle import hostcpu
    hostcpu.configure()          # fastest thread count <= affinity / cgroup quota (os.cpu_count() oversubscribes GPU boxes)
    cv_sd, mg_sd, cr_sd = state_dicts('full')
    cvc, mgc, crc = oracle_cfgs(FULL)
    video = weights.synthetic_video(2, 17, 256, 256, seed=0)
    with torch.no_grad():
        O.cvivit_tokenize(cv_sd, cvc, video)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.cpu_seconds:
            O.cvivit_tokenize(cv_sd, cvc, video)
            n += 1
        dt = time.perf_counter() - t0
    out = dict(value=n * 2 * 17 / dt, unit='frames/s', cores=torch.get_num_threads(), kind='port',
               sample=f'{n} calls of oracle cvivit_tokenize on (2,3,17,256,256) f32, {dt:.1f} s')
    if not args.no_sample:
        # sampler: 1 CFG MaskGit step + 1 CFG critic step at B = 1, scaled to the 18-step loop
        ctx = weights.synthetic_context(1, 12, 768, seed=1)
        ids = torch.full((1, 576), 65536)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.maskgit_cfg(mg_sd, mgc, ids, cond_scale=5., video_patch_shape=(9, 8, 8), context=ctx, text_mask=(ctx != 0).any(-1))
            t_mg = time.perf_counter() - t0
            t0 = time.perf_counter()
            O.critic_cfg(cr_sd, crc, ids, cond_scale=5., video_patch_shape=(9, 8, 8), context=ctx, text_mask=(ctx != 0).any(-1))
            t_cr = time.perf_counter() - t0
        est = 18 * t_mg + 17 * t_cr
        out['sample'] += f'; sampler: 1 CFG MaskGit forward {t_mg:.2f} s + 1 CFG critic forward {t_cr:.2f} s at B=1'
        out['sample_tokens_per_sec_estimate'] = 576 / est
    return out


def main():
    args = parse()
    torch.set_grad_enabled(False)
    from __graft_entry__ import build
    rank, local, ws = init_dist(args.gpus)
    if rank == 0:
        build()
    barrier_sync(ws)
    cv, mg, cr, ph = build_models(args.dtype, not args.no_sample)

    dt, used_graph, gemms, _ = bench_encode(cv, args, ws)
    frames = args.batch * 17 * args.steps * ws
    result = {
        'metric': 'cvivit_encode_frames_per_sec', 'value': frames / dt, 'unit': 'frames/s', 'n_gpus': ws,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype if args.dtype == 'bf16' else 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: C-ViViT (dim 512, patch 32, tpatch 2, depth 4+4, LFQ 65536) encode '
                               f'video -> token ids, ({args.batch},3,17,256,256) f32 per GPU resident in HBM',
                   'global_batch': args.batch * ws, 'frames_per_video': 17, 'parallelism': f'batch-shard x{ws}, no data-path collective',
                   'hip_graph': used_graph},
    }
    # roofline: dominant GEMM instantiation by total time
    if gemms:
        name, d = max(gemms.items(), key=lambda kv: kv[1]['seconds'])
        ach = d['flops'] / d['seconds'] / 1e12
        peak = PEAK_BF16_TFLOPS if args.dtype == 'bf16' else 157.3
        result['roofline'] = {'bound': 'mfma', 'kernel': name, 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                              'traffic': pmc_traffic(name, args), 'traffic_unit': 'HBM bytes per launch',
                              'traffic_source': 'profiles/gemm_hbm_pmc_r01.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, '
                                                'gfx950 x2 read correction); null when the run is not that configuration',
                              'launches_per_step': d['launches'],
                              'avg_launch_us': d['seconds'] / d['launches'] * 1e6,
                              'algorithmic_flops_per_launch': d['flops'] / d['launches'],
                              'all_gemm_variants': {k: {'launches': v['launches'], 'TFLOP/s': v['flops'] / v['seconds'] / 1e12,
                                                        'us_total': v['seconds'] * 1e6} for k, v in gemms.items()}}
    if not args.no_sample:
        result['sample'] = bench_sample(ph, args, ws)
        result['objective'] = bench_objective(ph, args, ws)
    if rank == 0 and ws == 1 and not args.no_cpu:
        result['cpu_baseline'] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(result))
    if ws > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
