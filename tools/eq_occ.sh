for ch in 0 256 320 384; do
export MX_EQ_SPEC_CHUNKS=$ch
[ $ch = 0 ] && unset MX_EQ_SPEC_CHUNKS
python bench.py --no-cpu-baseline --no-realtime --no-north-star --no-material-leg --no-scaling-probe --fir-ticks 0 --video-frames 0 --repeats 0 --steps 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks $ch:', round(d['ms_per_step'],3), d['roofline']['kernel_ms_per_step'], 'held', d['held_gates']['ms_per_step'], {k:v['ms_per_step'] for k,v in d['t_sweep'].items()})"
done
