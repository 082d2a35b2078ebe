# lateral fusions without torch.cat / .contiguous() copies in the training step (train_ops.join_channels): parity tests, then
# bench --mode train A/B on one box (train_ops._JOIN 1 | 0)
python -m pytest tests/test_gpu_bn_train.py tests/test_gpu_train_conv.py tests/test_gpu_train_step.py -x -q -m gpu 2>&1 | tail -3
for flag in 1 0 1 0; do
python - $flag <<'PY' 2>/dev/null | tail -1
import sys, runpy
sys.path.insert(0, ".")
import avtex.train_ops as t
t._JOIN = int(sys.argv[1])
sys.argv = ["bench.py", "--mode", "train", "--steps", "3", "--warmup", "2"]
print("JOIN", t._JOIN)
runpy.run_path("bench.py", run_name="__main__")
PY
done
