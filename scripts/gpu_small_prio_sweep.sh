for v in "-DQINCO_SMALL_PRIO2=2" "-DQINCO_SMALL_PRIO2=3" "-DQINCO_SMALL_PRIO3=1" "-DQINCO_SMALL_PRIO3=3"; do
  echo "##### $v"
  EXTRA="$v" scripts/gpu_small_timeline.sh "128 128 256 1 3 12288 7 2" "128 384 384 1 3 12288 2 3" 2>&1 | grep -E "===|launch 2|step 0 stamp  [37]"
done
