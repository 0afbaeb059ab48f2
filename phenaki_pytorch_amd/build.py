"""Build libphenaki_hip.so (gfx950) in-tree with hipcc.

    python -m phenaki_pytorch_amd.build [--force]

Cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', '_obj')
LIB = os.path.join(HERE, 'libphenaki_hip.so')
SOURCES = ['gemm.hip', 'qkv.hip', 'qkv_attn.hip', 'norm.hip', 'patch.hip', 'patch_embed.hip', 'elementwise.hip', 'attn.hip', 'sampler.hip', 'train.hip',
           'attn_train.hip', 'conv.hip', 'lfq_aux.hip']
HEADERS = ['common.hpp', 'gemm_core.hpp', 'gemm_dma.hpp', 'gemm_p8.hpp']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + os.environ.get('PK_EXTRA_HIPCC_FLAGS', '').split()


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace('.hip', '.o'))
        if force or _newer(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc()] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for src, rc, out in ex.map(compile_one, jobs):
                if verbose:
                    print(f'[build] {os.path.basename(src)}: {"ok" if rc == 0 else "FAILED"}')
                if rc != 0:
                    raise RuntimeError(f'hipcc failed on {src}:\n{out}')
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stdout + r.stderr)
        if verbose:
            print(f'[build] linked {LIB}')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
