# T = 64 submissions of the headline graph, one line per setting of the environment given in $1 ("VAR=a VAR=b ...")
FL="--steps 40 --warmup 5 --no-cpu-baseline --no-realtime --no-t-sweep --no-north-star --no-held-leg --no-material-leg --no-rate-leg --no-scaling-probe --fir-ticks 0 --repeats 0 --video-frames 0 --no-scaled-leg --no-contract-leg --ticks-per-step ${T:-64}"
for e in "X=0" "$@"; do
env $e python bench.py $FL | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', 'value', round(l['value']/1e6,1), 'ms/step', round(l['ms_per_step'],4), l['roofline'].get('kernel_ms_per_step'))"
done
