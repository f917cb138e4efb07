timeout 900 python -m pytest tests/test_gpu_eq_exact_spec.py tests/test_gpu_audio_parity.py tests/test_gpu_full_size.py tests/test_gpu_schedule.py tests/test_gpu_fusion.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-t-sweep --no-realtime --no-north-star --fir-ticks 0 --video-frames 0 --repeats 1 --steps 10 > gpurun_out/be.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/be.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d['eq_spec'], d['held_gates'] and d['held_gates']['ms_per_step'])
PY
