for n in 256 1024 2048 4096; do for b in 1 100000; do
  echo -n "strips=$n poles_below=$b: "
  MX_EQ_POLES_BELOW=$b timeout 300 python bench.py --strips $n --ticks-per-step 64 --no-cpu-baseline --fir-ticks 0 --video-frames 0 --repeats 0 --steps 1 --warmup 1 --no-held-leg --no-north-star 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["realtime"]["tick_us"])'
done; done
