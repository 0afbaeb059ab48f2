cd /root/repo
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "stats_handoff or dup_rows or layernorm_fold" 2>&1 | tail -4 > gpurun_out/t3.log
python bench.py --dtype bf16x3 --no-sample --no-cpu --no-parity-mode --legs decode > gpurun_out/bench_x3_k.json 2> gpurun_out/bench_x3_k.err
cp gpurun_out/bench_full.json gpurun_out/bench_x3_full.json
cat gpurun_out/t3.log
