#!/bin/bash
# (a) GAN leg after the pk_gemm routing of the 8-channel products; (b) per-kernel tables of the split-bf16 encode and sample legs
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gan_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k "step or identical" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu --no-parity-mode --no-kernels --no-sample --legs cvivit_gan_step > /dev/null 2> gpurun_out/gan_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_full.json'))
g=d.get('cvivit_gan_step'); print({k:g[k] for k in ('generator_step_ms','discriminator_step_ms','discriminator_step_no_penalty_ms','peak_memory_gb')})
PY
timeout 900 python bench.py --dtype bf16x3 --no-cpu --no-parity-mode --legs sample > /dev/null 2> gpurun_out/x3_bench.err; echo "bench rc=$?"
cp gpurun_out/bench_full.json gpurun_out/bench_x3_full.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_x3_full.json'))
print('encode', d['value'], d['ms_per_step'], 'sample', d['sample']['value'], d['sample'].get('ms'))
rows=d['kernels']
for leg in ('encode','sample'):
    rs=[r for r in rows if r['leg']==leg]
    tot=sum(r['us_total'] for r in rs)
    print('==', leg, 'total us', round(tot,1))
    for r in sorted(rs, key=lambda r:-r['us_total'])[:22]:
        print(f"  {r['us_total']:9.1f} us {100*r['us_total']/tot:5.1f}%  x{r['launches']:4d}  {r['avg_launch_us']:8.2f} us  frac {r.get('frac',0):.3f}  {r['kernel'][:90]}")
PY
