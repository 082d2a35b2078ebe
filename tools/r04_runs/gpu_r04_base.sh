mkdir -p gpurun_out/r04
python tools/probe_x3.py f16x3 83 > gpurun_out/r04/probe_b83_start.log 2>&1
python tools/probe_x3.py f16x3 166 > gpurun_out/r04/probe_b166_start.log 2>&1
python bench.py --no-cpu-baseline --no-train-leg --no-nxn-legs --no-precision-block --no-fast > gpurun_out/r04/bench_start.json 2> gpurun_out/r04/bench_start.err
tail -3 gpurun_out/r04/bench_start.json
