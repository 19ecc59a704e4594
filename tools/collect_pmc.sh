#!/bin/bash
# Hardware counters of the frame-path kernels (one rocprofv3 --pmc pass per counter group, each in its own run, kernel trace
# only next to it) for a workload given as "python3 <script> <args>": per kernel the mean of every counter per launch
# -> gpurun_out/<tag>/pmc.json, plus VGPR / SGPR / LDS / grid of every kernel from the kernel trace -> resources.json
#   tools/collect_pmc.sh <tag> <script> [args...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_ATOMIC_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  echo "pmc pass $i: $grp"
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed (see p$i.log)"
done
python3 tools/pmc_table.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
rm -rf $OUT/p[0-9]*/   # raw rocprofv3 output (tens of MB): only the summaries travel back
tail -5 $OUT/pmc_summary.txt
