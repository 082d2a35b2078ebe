# round 4, first batch: at-size e2e tests, BN group-0 running statistics, the bench line with the sharper inputs
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_e2e.py tests/test_gpu_bn_train.py tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r04/tests_a.log 2>&1
tail -5 gpurun_out/r04/tests_a.log
python bench.py --no-train-leg > gpurun_out/r04/bench_a.json 2> gpurun_out/r04/bench_a.err
tail -1 gpurun_out/r04/bench_a.json; grep -v "bench-detail" gpurun_out/r04/bench_a.err | tail -5
