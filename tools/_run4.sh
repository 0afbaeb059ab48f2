cd /root/repo
python -m pytest tests/test_train_gpu.py -m gpu -q -x -k "splitk" 2>&1 | tail -3 > gpurun_out/t4.log
python -m pytest tests/test_modules_gpu.py -m gpu -q -x -k "cvivit" 2>&1 | tail -3 >> gpurun_out/t4.log
python bench.py --dtype bf16x3 --legs sample --no-cpu --no-parity-mode > gpurun_out/bench_x3_s.json 2> gpurun_out/bench_x3_s.err
cp gpurun_out/bench_full.json gpurun_out/bench_x3_s_full.json
cat gpurun_out/t4.log
