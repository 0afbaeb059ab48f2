cd /root/repo
python bench.py --no-sample --no-cpu --no-parity-mode --legs none > gpurun_out/b6.json 2> gpurun_out/b6.err; cp gpurun_out/bench_full.json gpurun_out/bench_bf16_k.json
PK_PATCH_WIDE=0 python bench.py --no-sample --no-cpu --no-parity-mode --legs none > gpurun_out/b6o.json 2> gpurun_out/b6o.err; cp gpurun_out/bench_full.json gpurun_out/bench_bf16_k_old.json
tail -3 gpurun_out/b6.err | cut -c1-300
