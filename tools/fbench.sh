timeout 600 python -m pytest tests/test_gpu_fir_resample.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -4
for v in 0 1; do
MX_RESAMPLE_ONE_PER_LANE=$v python - <<'PY'
import sys, json, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, bench
torch.cuda.set_device(0); st = torch.cuda.Stream()
with torch.cuda.stream(st):
    r = bench.fir_leg(torch, st, 0, 128, 10, 2)
print(os.environ.get("MX_RESAMPLE_ONE_PER_LANE"), json.dumps({k: r[k] for k in ("ms_per_step", "kernel_ms_per_step", "resample_f64_tops")}))
PY
done
