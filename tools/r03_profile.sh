#!/bin/bash
# Round-3 profile of the bench command itself on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/r03_prof/   (per-kernel time; must agree with the bench line)
#   2. rocprofv3 --pmc <set> (one pass per set, no trace flags) -> gpurun_out/r03_pmc_<n>/
# then: python3 tools/r03_profile_summary.py  (writes profiles/r03_*).  The program follows `--` directly (no env / bash -c hop).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs"
rocprofv3 --kernel-trace --stats -d $OUT/r03_prof -o x --output-format csv -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $OUT/r03_bench_under_rocprof.json 2> $OUT/r03_prof.err
echo "kernel trace done"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d $OUT/r03_pmc_$i -o x --output-format csv -- $BENCH > $OUT/r03_pmc_$i.json 2> $OUT/r03_pmc_$i.err
  echo "pmc pass $i done: $line"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU
SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQC_TC_INST_REQ SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F
GRBM_GUI_ACTIVE
LIST
# the counters become profiles/r03_pmc_traffic.json ON THIS BOX first, so that the plain run below carries this run's own
# `traffic` (bench.py takes it only from a profile of the same kernel sources); the same summary is run again at home
cd $ROOT && rm -f $OUT/r03_bench.json && python3 tools/r03_profile_summary.py > /dev/null
python3 bench.py --steps 6 --warmup 2 > $OUT/r03_bench.json 2> $OUT/r03_bench.err
echo "plain bench done"
