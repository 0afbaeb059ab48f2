cd /root/repo
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/t2.log
python -m pytest tests/test_modules_gpu.py -m gpu -q -x 2>&1 | tail -15 >> gpurun_out/t2.log
python bench.py --dtype bf16x3 --legs sample --no-cpu --no-kernels > gpurun_out/bench_x3_new.json 2> gpurun_out/bench_x3_new.err
PK_LN_FOLD_X3=0 python bench.py --dtype bf16x3 --legs sample --no-cpu --no-kernels > gpurun_out/bench_x3_old.json 2> gpurun_out/bench_x3_old.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_x3 -o x3 -- python /root/repo/bench.py --dtype bf16x3 --encode-only --no-graph --groups 3 > /dev/null 2>&1
cd /root/repo; find gpurun_out/prof_x3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/x3_encode_kernel_stats.csv
rm -rf gpurun_out/prof_x3
tail -c 400 gpurun_out/bench_x3_new.json; tail -c 400 gpurun_out/bench_x3_old.json
