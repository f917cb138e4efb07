for w in 1280 1152 1024 896 768; do
  echo -n "W=$w "
  MX_EQ_SPEC_WARM=$w timeout 200 python bench.py --no-cpu-baseline --fir-ticks 0 --no-realtime --no-t-sweep --no-north-star --video-frames 0 --no-held-leg --repeats 0 --steps 20 --warmup 3 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["roofline"]["kernel_ms_per_step"], d["eq_spec"])'
done
