#!/bin/bash
# One counter pass for instruction fetch + one for the effective clock over the bench command; the program
# follows `--` directly.  Output: gpurun_out/r03_pmc_ic/, gpurun_out/r03_pmc_clk/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES -d $OUT/r03_pmc_ic -o x --output-format csv -- $BENCH > $OUT/r03_pmc_ic.json 2> $OUT/r03_pmc_ic.err
echo "icache pass done"
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/r03_pmc_clk -o x --output-format csv -- $BENCH > $OUT/r03_pmc_clk.json 2> $OUT/r03_pmc_clk.err
echo "clock pass done"
