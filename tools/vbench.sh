timeout 900 python -m pytest tests/test_gpu_video_parity.py tests/test_gpu_video_graph.py tests/test_gpu_monitor_sink.py tests/test_gpu_ingest.py tests/test_fastdiv.py -x -q 2>&1 | tail -15 > gpurun_out/tv.log
timeout 300 python bench.py --strips 64 --ticks-per-step 64 --steps 2 --no-cpu-baseline --no-t-sweep --no-realtime --no-north-star --fir-ticks 0 --no-held-leg --repeats 0 > gpurun_out/bv.log 2>&1
tail -6 gpurun_out/tv.log; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bv.log').read().strip().splitlines()[-1]); v=d['video']; print({k:v[k] for k in ('value','device_us_per_frame','hbm_frac_moved_bytes_device')})
except Exception as e: print('ERR',e); print(open('gpurun_out/bv.log').read()[-1500:])
PY
