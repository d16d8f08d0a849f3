#!/bin/bash
# (EXTRA="-DQINCO_SMALL_FORCE_SINGLE ..." adds compile flags)
# Timeline of the small-launch decode kernel (scripts/ubench/small_timeline.hip) on the GPU box.  usage: gpu_small_timeline.sh "D DE DH F2 NT R steps L" ...
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  hipcc -O3 -std=c++20 --offload-arch=gfx950 $EXTRA -DQINCO_EXPERIMENT -DQINCO_TIMELINE -DQD=$1 -DQDE=$2 -DQDH=$3 -DQF2=$4 -DQNT=$5 -Iqinco_amd/csrc \
        scripts/ubench/small_timeline.hip -o /tmp/small_timeline_$1_$2_$3_$5 || exit 1
  echo "=== D=$1 De=$2 Dh=$3 FOLD2=$4 NT=$5 R=$6 steps=$7 L=$8"
  /tmp/small_timeline_$1_$2_$3_$5 $6 $7 $8
done
