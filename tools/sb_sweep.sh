for sb in 16 32; do
  echo -n "SB=$sb "
  MX_EQ_SPEC_SB=$sb timeout 300 python bench.py --no-cpu-baseline --fir-ticks 0 --no-realtime --no-t-sweep --no-north-star --video-frames 0 --no-held-leg --repeats 0 --steps 10 --warmup 2 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["roofline"]["kernel_ms_per_step"])'
done
