# training: pointwise layers (forward / stride-1 input gradient) on the streaming fp32-in / fp32-out kernel (avt_pw_x3_f32).  Parity tests, the
# layer alone (tools/conv_layer_bench.py IO32=fwd), then bench --mode train with train_ops._PW_F32 = 1 | 0 alternating on one box
python -m pytest tests/test_gpu_train_conv.py tests/test_gpu_train_step.py tests/test_gpu_bn_train.py -x -q -m gpu 2>&1 | tail -2
for sh in "128 512 1 1 1 128 8 28 28" "64 256 1 1 1 128 8 56 56" "256 1024 1 1 1 128 8 14 14" "8 32 1 1 1 128 32 56 56"; do IO32=fwd python tools/conv_layer_bench.py $sh 2>&1 | tail -1; done
for flag in 1 0 1 0 1 0; do
python - $flag <<'PY' 2>/dev/null | tail -1 | cut -c125-215
import sys, runpy
sys.path.insert(0, ".")
import avtex.train_ops as t
t._PW_F32 = int(sys.argv[1])
sys.argv = ["bench.py", "--mode", "train", "--steps", "3", "--warmup", "2"]
print("PW_F32", t._PW_F32)
runpy.run_path("bench.py", run_name="__main__")
PY
done
