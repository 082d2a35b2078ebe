#!/bin/bash
# the whole GPU suite + smoke, then the default bench run (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/r05_suite_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r05_smoke.log
timeout 900 python tools/probe_wgrad_xl.py 120 > gpurun_out/r05_probe_wgrad_xl.log 2>&1
python bench.py --steps ${1:-8} --warmup 2 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
cp bench_detail.json gpurun_out/r05_bench_detail.json
tail -c 3500 gpurun_out/r05_bench.json
