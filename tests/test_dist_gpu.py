"""The batch-sharded sampler on the REAL product path under an initialised process group: two ranks, each running
the real Phenaki.sample on its shard and the one all-gather of the decoded videos -- one rank per device when two devices are
visible (then the `nccl` variant is RCCL over xGMI and is NOT skipped), both ranks on the one GPU otherwise (RCCL refuses two ranks
on one device: the gloo variant covers the same path).  Needs a real MI355X (-m gpu).  The 8-GPU scaling run is the driver's;
this pins correctness of the path it times."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, ws, port, backend, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.set_grad_enabled(False)
    dev = rank % max(1, torch.cuda.device_count())              # one rank per device when there are two; else both share the one GPU
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group(backend, rank=rank, world_size=ws, device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=ws)
    from oracle import weights
    from oracle.configs import TINY
    from tests.util import load_product
    import phenaki_pytorch_amd as P
    _, _, _, ph = load_product('tiny', TINY, device=f'cuda:{dev}')
    ctx_row = weights.synthetic_context(1, 6, TINY['maskgit']['dim_context'], seed=2).cuda()
    ph.encode_texts = lambda texts, output_device=None: ctx_row.expand(len(texts), -1, -1).contiguous()   # IDENTICAL prompts
    texts = ['p'] * 5                                           # 5 items over 2 ranks: shards of 3 and 2 (ragged tail padded + trimmed)
    lo, hi = P.shard_batch(len(texts))
    assert (lo, hi) == ((0, 3) if rank == 0 else (3, 5))

    torch.manual_seed(5)                                        # the SAME torch seed on every rank
    local, local_ids = ph.sample(texts=texts[lo:hi], num_frames=5, cond_scale=5., _return_ids=True)
    seed_used = ph._pk_last_seed
    torch.manual_seed(5)
    full = P.sample_sharded(ph, texts=texts, num_frames=5, cond_scale=5.)
    assert tuple(full.shape) == (5, 3, 5, 64, 64) and torch.isfinite(full).all()
    assert torch.equal(full[lo:hi], local), 'gathered order: my shard is not where shard_batch says it is'
    torch.save(dict(full=full.cpu(), local=local.cpu(), ids=local_ids.cpu(), seed=seed_used, lo=lo, hi=hi), os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()
    # the same shard in a single process (no process group) with the rank's seed: bit-identical
    again, again_ids = ph.sample(texts=texts[lo:hi], num_frames=5, cond_scale=5., _return_ids=True, _seed=seed_used)
    assert torch.equal(again_ids, local_ids) and torch.equal(again, local), 'shard differs from the single-process run with the same seed'
    # and through the captured hipGraph: same ids for the same seed
    ph.enable_sample_graph(True)
    g_vid, g_ids = ph.sample(texts=texts[lo:hi], num_frames=5, cond_scale=5., _return_ids=True, _seed=seed_used)
    g_vid2, g_ids2 = ph.sample(texts=texts[lo:hi], num_frames=5, cond_scale=5., _return_ids=True, _seed=seed_used)      # replay
    assert torch.equal(g_ids.cpu(), local_ids.cpu()) and torch.equal(g_ids2.cpu(), local_ids.cpu()), 'hipGraph sampling differs from eager'
    assert torch.equal(g_vid2, local)


@pytest.mark.parametrize('backend', ['nccl', 'gloo'])
def test_sample_sharded_two_ranks_real_path(tmp_path, backend):
    """(i) gathered order, (ii) per-rank noise streams differ under a common torch seed, (iii) each shard equals the
    single-process result for its seed (eager and hipGraph)."""
    ws, port = 2, _free_port()
    two_devices = torch.cuda.device_count() >= 2
    try:
        mp.spawn(_rank_main, args=(ws, port, backend, str(tmp_path)), nprocs=ws, join=True)
    except Exception as e:                                       # noqa: BLE001
        # only on a ONE-device box may the RCCL variant be skipped (RCCL refuses two ranks on one device); with two devices it must run
        if backend == 'nccl' and not two_devices and any(k in str(e) for k in ('Duplicate GPU', 'duplicate', 'invalid usage', 'unhandled system error', 'NCCL', 'RCCL')):
            pytest.skip(f'one visible device and RCCL refuses two ranks on it ({str(e)[:120]}...): covered by the gloo variant on the same path')
        raise
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f'rank{r}.pt'), weights_only=False) for r in range(2))
    assert torch.equal(r0['full'], r1['full']), 'ranks disagree on the gathered batch'
    assert torch.equal(r0['full'][3:5], r1['local']) and torch.equal(r1['full'][0:3], r0['local'])
    assert r0['seed'] != r1['seed'], 'ranks share a noise stream'
    # identical prompts and torch seed: identical noise streams would make rank 1's first two videos equal rank 0's
    assert not torch.equal(r0['ids'][:2], r1['ids'][:2]), 'per-rank noise streams must differ'


def test_bench_gpus2_launches_two_ranks_by_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher must start 2 ranks itself and report n_gpus = 2 with the collective
    that moved the videos (RCCL when two devices are visible; on a one-device box the PK_BENCH_ONE_DEVICE dry run over gloo)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    two = torch.cuda.device_count() >= 2
    if not two:
        env['PK_BENCH_ONE_DEVICE'] = '1'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--groups', '2', '--batch', '2',
                        '--sample-batch', '1', '--no-cpu', '--no-parity-mode', '--no-kernels', '--no-graph', '--legs', 'sample'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4
    s = d['sample']
    assert s['global_batch'] == 2 and s['all_gather_us'] > 0
    assert s['backend'] == ('nccl' if two else 'gloo')
    if two:
        assert s['nccl_version'] and 'RCCL' in s['collective']
    # and without enough devices / the dry-run switch it fails loudly instead of running one rank
    if not two:
        env.pop('PK_BENCH_ONE_DEVICE')
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--encode-only'], env=env, capture_output=True, text=True, timeout=300)
        assert r2.returncode != 0 and 'HIP device(s) visible' in (r2.stderr + r2.stdout)


def test_bench_gpus8_whole_flow_on_one_device(tmp_path):
    """VERDICT r5 #6a: the driver's first real SCALE run is `bench.py --gpus 8`; no 8-GPU node was ever available here, so the rank-count paths of
    that flow (spawn of 8 ranks, shard arithmetic at 8, configs[3] = 32 videos -> 4 per rank, configs[4] = 8 videos -> 1 per rank, the all-gather of
    8 shards, max-over-ranks timing, rank 0's JSON line) are executed as 8 processes on ONE device over gloo (PK_BENCH_ONE_DEVICE)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, PK_BENCH_ONE_DEVICE='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--groups', '2', '--batch', '1',
                        '--sample-batch', '4', '--no-cpu', '--no-parity-mode', '--no-kernels', '--no-graph', '--legs', 'sample,make_video'],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])          # the compact line the driver parses
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 8 and d['scaling'] == 'weak'
    s = d['sample']
    assert s['global_batch'] == 32 and s['batch_per_gpu'] == 4 and s['backend'] == 'gloo' and s['all_gather_us'] > 0
    assert abs(s['value'] - 32 * 576 / (s['ms'] * 1e-3)) < 2e-3 * s['value']
    full = json.load(open(os.path.join(ROOT, 'gpurun_out', 'bench_full.json')))                # rank 0's full report
    mv = full['make_video']
    assert mv['videos'] == 8 and mv['videos_per_gpu'] == 1 and mv['value'] > 0
    assert 'scaling_projection' not in full, 'a projection belongs to the 1-GPU report only'
    with open(os.path.join(ROOT, 'gpurun_out', 'bench_gpus8_one_device.json'), 'w') as f:
        json.dump(dict(compact_line=d, sample=full['sample'], make_video=mv), f, indent=1)


def _rccl_one_rank(rank, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from oracle import weights
    from oracle.configs import TINY
    from tests.util import load_product
    import phenaki_pytorch_amd as P
    torch.set_grad_enabled(False)
    _, mg, _, ph = load_product('tiny', TINY, device='cuda:0')
    ctx_row = weights.synthetic_context(1, 6, TINY['maskgit']['dim_context'], seed=2).cuda()
    ph.encode_texts = lambda texts, output_device=None: ctx_row.expand(len(texts), -1, -1).contiguous()
    torch.manual_seed(5)
    full = P.sample_sharded(ph, texts=['p'] * 3, num_frames=5, cond_scale=5., _force_collective=True)     # all_gather_into_tensor through RCCL
    seed_used = ph._pk_last_seed
    direct = ph.sample(texts=['p'] * 3, num_frames=5, cond_scale=5., _seed=seed_used)
    assert torch.equal(full, direct)
    params = [torch.nn.Parameter(torch.randn(300, 70, device='cuda')), torch.nn.Parameter(torch.randn(33, device='cuda'))]
    for p in params:
        p.grad = torch.randn_like(p)
    before = [p.grad.clone() for p in params]
    assert P.all_reduce_gradients(params, bucket_mb=0.05, force=True) == 2             # two buckets, all_reduce through RCCL, averaged over 1 rank
    assert all(torch.equal(p.grad, b) for p, b in zip(params, before))
    # the overlapped reducer on RCCL: async all_reduce issued from the autograd thread's hooks during a real backward, waited in finish()
    for m in (mg,):
        m.train()
    torch.set_grad_enabled(True)
    mparams = list(mg.parameters())
    g = torch.Generator().manual_seed(40)
    ids = torch.randint(0, TINY['maskgit']['num_tokens'], (2, 3, 4, 4), generator=g).cuda()
    ctx = weights.synthetic_context(2, 6, TINY['maskgit']['dim_context'], seed=3).cuda()
    draws = dict(rand_step=torch.tensor([1, 4]), perm_noise=weights.uniform_noise((2, 48), 750), gumbel_u=weights.uniform_noise((2, 48, TINY['maskgit']['num_tokens']), 760))
    ph(video_codebook_ids=ids, text_embeds=ctx, _draws=draws, only_train_generator=True).backward()
    plain = [p.grad.clone() if p.grad is not None else None for p in mparams]
    for p in mparams:
        p.grad = None
    reducer = P.GradientReducer(mparams, bucket_mb=0.25, force=True)
    ph(video_codebook_ids=ids, text_embeds=ctx, _draws=draws, only_train_generator=True).backward()
    in_backward = reducer.collectives
    n_coll = reducer.finish()
    assert in_backward >= 1 and n_coll == len(reducer.buckets) >= 2
    # (a parameter that got no gradient ends with the reduced ZEROS as its .grad -- every replica then steps it alike, dist.py / ADVICE r4)
    assert all((p.grad is None or not p.grad.any()) if a is None else torch.allclose(a, p.grad, rtol=1e-4, atol=1e-7) for a, p in zip(plain, mparams)), \
        'a one-rank average must return the gradients unchanged (embedding rows are float atomics: not bit-equal run to run)'
    torch.save(dict(ok=True, version=torch.cuda.nccl.version()), os.path.join(out_dir, 'rccl.pt'))
    dist.destroy_process_group()


def test_rccl_executes_the_collectives_on_this_box(tmp_path):
    """a ONE-rank RCCL communicator (the most a single-GPU box allows) runs the two collectives of the package on device tensors -- the
    sampler's all_gather_into_tensor, the training step's bucketed all_reduce and the overlapped GradientReducer (async all_reduce from the
    autograd hooks of a real backward) -- so the RCCL code path itself is exercised here, not
    only gloo's; the multi-rank variants above cover ordering and per-rank streams"""
    mp.spawn(_rccl_one_rank, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = torch.load(os.path.join(str(tmp_path), 'rccl.pt'), weights_only=False)
    assert r['ok'] and r['version']


def _ddp_rank(rank, ws, port, out_dir, overlapped):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dev = rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    two = torch.cuda.device_count() >= 2
    if two:
        dist.init_process_group('nccl', rank=rank, world_size=ws, device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=ws)
    from oracle import weights
    from oracle.configs import TINY
    from tests.util import load_product
    import phenaki_pytorch_amd as P
    _, mg, cr, ph = load_product('tiny', TINY, device=f'cuda:{dev}')
    params = [p for p in list(mg.parameters()) + list(cr.parameters())]
    g = torch.Generator().manual_seed(40)
    ids = torch.randint(0, TINY['maskgit']['num_tokens'], (4, 3, 4, 4), generator=g)
    ctx = weights.synthetic_context(4, 6, TINY['maskgit']['dim_context'], seed=3, pad_last=1)
    n = 48

    def draws_of(r):
        return dict(rand_step=torch.tensor([1 + r, 4]), perm_noise=weights.uniform_noise((2, n), 750 + r),
                    gumbel_u=weights.uniform_noise((2, n, TINY['maskgit']['num_tokens']), 760 + r))

    lo = 2 * rank
    reducer = P.GradientReducer(params, bucket_mb=0.25) if overlapped else None     # hooks: buckets go out while backward is still running
    with torch.enable_grad():
        loss = ph(video_codebook_ids=ids[lo:lo + 2].to(f'cuda:{dev}'), text_embeds=ctx[lo:lo + 2].to(f'cuda:{dev}'), _draws=draws_of(rank))
        loss.backward()
    if overlapped:
        started_in_backward = reducer.collectives
        n_coll = reducer.finish()
        assert started_in_backward >= 1 and n_coll == len(reducer.buckets)
        reducer.remove()
    else:
        n_coll = P.all_reduce_gradients(params, bucket_mb=0.25)
    assert n_coll >= 2
    torch.save(dict(loss=float(loss.detach()), grads={i: p.grad.cpu() for i, p in enumerate(params) if p.grad is not None and p.numel()}),
               os.path.join(out_dir, f'ddp{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # the same two half-batches in ONE process, gradients accumulated with the 1 / world_size weight DDP's averaging implies
        for p in params:
            p.grad = None
        with torch.enable_grad():
            for r in range(ws):
                l = ph(video_codebook_ids=ids[2 * r:2 * r + 2].to(f'cuda:{dev}'), text_embeds=ctx[2 * r:2 * r + 2].to(f'cuda:{dev}'), _draws=draws_of(r))
                (l / ws).backward()
        torch.save({i: p.grad.cpu() for i, p in enumerate(params) if p.grad is not None and p.numel()}, os.path.join(out_dir, 'single.pt'))


@pytest.mark.parametrize('overlapped', [False, True])
def test_data_parallel_training_step_equals_single_process_accumulation(tmp_path, overlapped):
    """SURVEY.md 8f row 3 on the real path: two ranks each run Phenaki.forward + backward on their half of the batch and average the
    gradients with the bucketed all-reduce (RCCL across two devices when there are two, gloo with both ranks on the one GPU otherwise);
    every rank must end with the gradients one process gets by accumulating the two half-batch losses with weight 1/2.  overlapped:
    the same through GradientReducer, whose hooks start each bucket's all-reduce during backward"""
    ws, port = 2, _free_port()
    mp.spawn(_ddp_rank, args=(ws, port, str(tmp_path), overlapped), nprocs=ws, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f'ddp{r}.pt'), weights_only=False) for r in range(2))
    single = torch.load(os.path.join(str(tmp_path), 'single.pt'), weights_only=False)
    assert r0['loss'] != r1['loss'], 'the ranks must have seen different half-batches'
    # a parameter NO rank produced a gradient for keeps .grad = None under the reducer too (used-flag all-reduce), so the sets are equal
    assert r0['grads'].keys() == r1['grads'].keys() == single.keys()
    for i, ref in single.items():
        assert torch.equal(r0['grads'][i], r1['grads'][i]), 'ranks disagree after the all-reduce'
        scale = float(ref.abs().max())
        if scale > 1e-6:                                    # (the structurally zero position-bias bias is noise on both sides)
            assert float((r0['grads'][i] - ref).abs().max()) <= 1e-4 * scale + 1e-7, f'parameter {i}'
