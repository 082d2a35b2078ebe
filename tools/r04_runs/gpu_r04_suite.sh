# the whole GPU suite (no -x: report every failure)
mkdir -p gpurun_out/r04
python -m pytest tests -q -m gpu > gpurun_out/r04/tests_full.log 2>&1
tail -6 gpurun_out/r04/tests_full.log
