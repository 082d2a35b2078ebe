# the whole GPU suite + a bench run after the hygiene commit
mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu > gpurun_out/r04/tests_full.log 2>&1
tail -4 gpurun_out/r04/tests_full.log
python bench.py --no-train-leg --no-cpu-baseline > gpurun_out/r04/bench_b.json 2> gpurun_out/r04/bench_b.err
tail -1 gpurun_out/r04/bench_b.json | cut -c1-400
python bench.py --no-train-leg --no-cpu-baseline --no-fast --no-nxn-legs --no-precision-block --inputs r03 2>/dev/null | tail -1 | cut -c1-300
