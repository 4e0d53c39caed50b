#!/bin/bash
# Round over round on ONE box: the round-2 tree against this tree, alternating.  -> gpurun_out/ab_round2.txt
# Set-up (build container): mkdir .ab_r2 && git archive b705984 | tar -x -C .ab_r2 && make -C .ab_r2/nerf-workspaces-explorer_amd/csrc -j8
# (.ab_r2/ is git-ignored; it travels to the GPU box with the snapshot: its own library, its own bench.py).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_round2.txt
mkdir -p $ROOT/gpurun_out; : > $OUT
for rep in 1 2 3; do
  (cd $ROOT/.ab_r2 && python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('round 2 rep $rep: kernel_ms %.2f ms_per_step %.2f' % (d['roofline']['kernel_ms'], d['ms_per_step']))" | tee -a $OUT
  (cd $ROOT && python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null) | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('round 3 rep $rep: kernel_ms %.2f ms_per_step %.2f' % (d['roofline']['kernel_ms'], d['ms_per_step']))" | tee -a $OUT
done
