#!/bin/bash
# Round-2 profile of the bench command itself on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/r02_prof/   (per-kernel time; must agree with the bench line)
#   2. rocprofv3 --pmc <set> (one pass per set, no trace flags) -> gpurun_out/r02_pmc_<n>/
# then: python3 tools/r02_profile_summary.py  (writes profiles/r02_*).  The program follows `--` directly (no env / bash -c hop).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/r02_prof -o x --output-format csv -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/r02_bench_under_rocprof.json 2> $OUT/r02_prof.err
echo "kernel trace done"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line -d $OUT/r02_pmc_$i -o x --output-format csv -- $BENCH > $OUT/r02_pmc_$i.json 2> $OUT/r02_pmc_$i.err
  echo "pmc pass $i done: $line"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU
SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
LIST
cd $ROOT && python3 bench.py --steps 6 --warmup 2 > $OUT/r02_bench.json 2> $OUT/r02_bench.err
echo "plain bench done"
