cd /root/repo
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "patch_embed" 2>&1 | tail -6 > gpurun_out/t5.log
python -m pytest tests/test_modules_gpu.py -m gpu -q -x -k "cvivit or bf16_blocks or tokenize or encode" 2>&1 | tail -6 >> gpurun_out/t5.log
for w in 1 0 1 0; do PK_PATCH_WIDE=$w python bench.py --encode-only --groups 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide=$w', d['value'], d['ms_per_step'])" >> gpurun_out/t5.log; done
python bench.py --no-sample --no-cpu --no-parity-mode --legs none > /dev/null 2> gpurun_out/b5.err; cp gpurun_out/bench_full.json gpurun_out/bench_bf16_k.json
cat gpurun_out/t5.log
