#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MX_TAIL_GATE=1
echo "under rocprof:"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_x -- python $R/tools/eq_sweep.py --toggle --steps 20 --no-profile --ticks 256 --overlap-tail 2>&1 | grep strips
echo "plain:"; python $R/tools/eq_sweep.py --toggle --steps 20 --no-profile --ticks 256 --overlap-tail 2>&1 | grep strips
echo "plain, AMD_SERIALIZE_KERNEL unset, HIP_FORCE_DEV_KERNARG=1:"; HIP_FORCE_DEV_KERNARG=1 python $R/tools/eq_sweep.py --toggle --steps 20 --no-profile --ticks 256 --overlap-tail 2>&1 | grep strips
echo "GPU_MAX_HW_QUEUES=2:"; GPU_MAX_HW_QUEUES=2 python $R/tools/eq_sweep.py --toggle --steps 20 --no-profile --ticks 256 --overlap-tail 2>&1 | grep strips
echo "GPU_MAX_HW_QUEUES=8:"; GPU_MAX_HW_QUEUES=8 python $R/tools/eq_sweep.py --toggle --steps 20 --no-profile --ticks 256 --overlap-tail 2>&1 | grep strips
