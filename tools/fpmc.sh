# SQ counter passes of the FIR + resampler leg alone:  gpurun -- 'bash tools/fpmc.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/fp; timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/fp -- python $R/tools/fleg.py 128 6 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/fp -name "*counter_collection.csv" | head -1) | grep -A9 -E "k_resample|k_fir\(" | grep -v history
done
# the clock the chip sustains under these kernels
rm -rf /tmp/fc; timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/fc -- python $R/tools/fleg.py 128 10 > /dev/null 2>&1
python $R/tools/pmc_clock.py $(find /tmp/fc -name "*counter_collection.csv" | head -1) $(find /tmp/fc -name "*kernel_trace.csv" | head -1) /tmp/fclock.json
