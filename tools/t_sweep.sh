for t in 1024 2048 4096; do
  echo -n "T=$t "
  timeout 300 python bench.py --ticks-per-step $t --no-cpu-baseline --fir-ticks 0 --no-realtime --no-t-sweep --no-north-star --video-frames 0 --no-held-leg --repeats 0 --steps 10 --warmup 2 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["value"]/1e6,1), d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["eq_spec"])'
done
