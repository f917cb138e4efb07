#!/bin/bash
# The driver's N = 8 launch line at FULL size -- python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 -- on a ONE-GPU box: every rank on GPU 0
# (MX_BENCH_SHARE_GPU), torch.distributed on gloo, the library's exchange on the RCCL test double (tests/helpers/fake_rccl.c).  Times mean nothing (eight ranks share the chip and
# the double stages through host memory); what it shows is that the job as the driver will run it -- 128 strips x 16 384 ticks per rank and step, ~20 GB of ports per rank, the
# slices exchange of 105 MB buses, both tick policies, the oracle replay on rank 0 -- builds, fits and checks out.   gpurun -- 'bash tools/n8_one_gpu.sh [steps] [warmup]'
cd $GRAFT_REPO_ROOT
gcc -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/libfake_rccl.so tests/helpers/fake_rccl.c -L/opt/rocm/lib -lamdhip64 -lrt || exit 1
S=$(date +%s)
FAKE_RCCL_CAP_MB=512 MX_BENCH_SHARE_GPU=1 MX_BENCH_DIST_BACKEND=gloo MX_RCCL_LIB=/tmp/libfake_rccl.so timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
  --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 8 --steps ${1:-5} --warmup ${2:-2} --full-out /tmp/full8.json > /tmp/line8.json 2> /tmp/err8.log
echo "rc=$? wall=$(( $(date +%s) - S ))s line=$(wc -c < /tmp/line8.json) bytes"
grep -v Gloo /tmp/err8.log | tail -6 | cut -c1-300
cat /tmp/line8.json
python - <<'PY'
import json
d = json.load(open('/tmp/full8.json'))
print("headline_parity", d['headline_parity']['verdict'], d['headline_parity']['buses']['verdict'], d['headline_parity'].get('seconds'), "s")
print("exchange", d['exchange']['mode'], d['exchange']['parity_check'])
print("other policy", d['other_policy']['ticks_per_step'], d['other_policy']['parity'])
PY
