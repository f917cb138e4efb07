for cfg in "MX_RESAMPLE_PW=0" "MX_RESAMPLE_PW=1 MX_RESAMPLE_PW_WAVES=4" "MX_RESAMPLE_PW=1 MX_RESAMPLE_PW_WAVES=5" "MX_RESAMPLE_PW=1 MX_RESAMPLE_PW_WAVES=6" "MX_RESAMPLE_PW=1"; do
  env $cfg python tools/fleg.py 128 10 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', {k: round(v,4) for k,v in l['kernel_ms_per_step'].items()}, 'fc', {k: round(v,4) for k,v in l['fp_contract']['kernel_ms_per_step'].items()})"
done
