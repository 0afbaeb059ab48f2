cd /root/repo
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "patch_embed" 2>&1 | tail -3
for st in 1 0; do
PK_PATCH_STAGGER=$st python bench.py --no-sample --no-cpu --no-parity-mode --legs none > gpurun_out/b8.json 2> gpurun_out/b8.err; cp gpurun_out/bench_full.json gpurun_out/bench_bf16_st$st.json
done
