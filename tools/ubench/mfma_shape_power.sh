#!/bin/bash
# built on the GPU box: hipcc is there
set -e
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_shape_power.hip -o /tmp/mfma_shape_power
for args in "32 0" "16 0" "32 50" "16 50"; do /tmp/mfma_shape_power $args | tail -2; done
