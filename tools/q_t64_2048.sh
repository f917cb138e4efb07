FL="--steps 40 --warmup 5 --no-cpu-baseline --no-realtime --no-t-sweep --no-north-star --no-held-leg --no-material-leg --no-rate-leg --no-scaling-probe --fir-ticks 0 --repeats 0 --video-frames 0 --no-scaled-leg"
for T in 64 2048; do
S=40; [ $T = 2048 ] && S=10
python bench.py $FL --steps $S --ticks-per-step $T | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T', $T, 'value', round(l['value']/1e6,1), 'ms/step', round(l['ms_per_step'],4), l['roofline'].get('kernel_ms_per_step'), 'fc', l.get('fp_contract',{}).get('kernel_ms_per_step'), l['eq_spec'])"
done
