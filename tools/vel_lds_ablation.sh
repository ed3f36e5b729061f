#!/bin/bash
# GPU box: upper bound of staging the record's u/v patches through LDS (diagnostic ablations, WRONG results by design).
#   vglob : uniform drift, the four velocity loads kept as dependencies (global memory, as shipped)
#   vlds  : the same with four LDS reads at a cell-dependent address in their place
#   dtree : the shipped loop in a diagnostic build (reference point for the cost of the ablation's scaffolding)
# build:  tools/build_variant.sh vglob -DSITRK_DIAG -DSITRK_NO_STAMPS -DSITRK_ABL_VELCONST
#         tools/build_variant.sh vlds  -DSITRK_DIAG -DSITRK_NO_STAMPS -DSITRK_ABL_VELCONST -DSITRK_ABL_VELLDS
#         tools/build_variant.sh dtree -DSITRK_DIAG -DSITRK_NO_STAMPS
TAG=${1:-r04au}
tools/ab_libs_n.sh ${TAG}_c3 3 "dtree vglob vlds" --steps 2048 --warmup 64 --no-cpu-baseline --no-c2 --only-fused
tools/ab_libs_n.sh ${TAG}_c2 3 "dtree vglob vlds" --config c2 --steps 512 --warmup 64 --no-cpu-baseline --no-c2 --only-fused
