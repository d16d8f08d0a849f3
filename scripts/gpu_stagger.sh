#!/bin/bash
# start offset between the two residents of a CU (experiment builds, scripts/build_exp_lib.py stagN -DQINCO_STAGGER_EXP=N)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
for n in 0 4 8 12; do
  lib=scripts/exp_libs/lib_stag$n.so; [ $n = 0 ] && lib=qinco_amd/libqinco_hip.so
  [ -f $lib ] || continue
  echo "== stagger $n x 8128 cycles"
  QINCO_HIP_LIB=$lib python scripts/exp_khead_ab.py S --reps 3 2>&1 | grep vec_per_s
done | tee $O/stagger.log
