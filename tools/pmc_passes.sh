#!/bin/bash
# Counter passes over tools/prof_run.py (one render launch, one workgroup per CU); output under gpurun_out/pmc_<n>.
# Usage on the GPU box: bash tools/pmc_passes.sh   (then: python3 tools/pmc_summary.py gpurun_out/pmc_*)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  rocprofv3 --pmc $line -d $ROOT/gpurun_out/pmc_$i -o x --output-format csv -- python3 $ROOT/tools/prof_run.py > $ROOT/gpurun_out/pmc_$i.log 2>&1
  echo "pass $i done: $line"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL
SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT
LIST
