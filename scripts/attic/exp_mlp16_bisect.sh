#!/bin/bash
# the 16-row kernel at 3 workgroups per CU (pad 0) under each bisection switch: rows that differ run to run
for b in ${@:-0 1 2 4 8 16 64 128}; do
  if [ $b = 0 ]; then unset QINCO_HIP_LIB; else export QINCO_HIP_LIB=$PWD/scripts/exp_libs/lib_e16_$b.so; fi
  echo "== QINCO_EXP16=$b"
  timeout 300 python scripts/exp_coresidency.py 0 2>/dev/null | grep '"48,196"' | cut -c1-200
  timeout 300 python scripts/exp_coresidency.py 0 2>/dev/null | grep '"48,196"' | cut -c1-200
done
