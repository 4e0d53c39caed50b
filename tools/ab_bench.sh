#!/bin/bash
# A/B timing on ONE box: the product library against variant builds (make -C csrc variant NAME=.. VFLAGS=..), alternating runs so
# that the drift of the die is shared.  Usage (on the GPU box): bash tools/ab_bench.sh name1 [name2 ...] -> gpurun_out/ab_<names>.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_$(echo "$@" | tr ' ' '_').txt
mkdir -p $ROOT/gpurun_out
: > $OUT
for rep in 1 2 3; do
  for v in product "$@"; do
    if [ "$v" = product ]; then unset NWE_LIB; else export NWE_LIB=$ROOT/nerf-workspaces-explorer_amd/csrc/exp/libnwe_$v.so; fi
    python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v rep $rep: kernel_ms %.2f ms_per_step %.2f' % (d['roofline']['kernel_ms'], d['ms_per_step']))" | tee -a $OUT
  done
done
