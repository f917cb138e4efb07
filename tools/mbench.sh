timeout 600 python -m pytest tests/test_gpu_eq_exact_spec.py -x -q 2>&1 | tail -3
timeout 400 python bench.py --no-cpu-baseline --no-t-sweep --no-realtime --no-north-star --fir-ticks 0 --video-frames 0 --repeats 0 --steps 6 --no-held-leg > gpurun_out/bm.log 2>&1
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bm.log').read().strip().splitlines()[-1])
    print(d['ms_per_step'], json.dumps(d['material'], indent=1))
except Exception as e:
    print("ERR", e); print(open('gpurun_out/bm.log').read()[-2000:])
PY
