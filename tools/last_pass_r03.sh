#!/bin/bash
# GPU box: the round's last pass over the final tree - soaks and fuzzers with seeds not used before.
# PART=1|2|3 picks a third (a gpurun call is at most 20 minutes). Every step under its own time
# limit; the script stops at the first failure.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/last_pass
mkdir -p $OUT
cd $R
run() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a $OUT/summary.txt
  timeout -k 10 $secs "$@" > $OUT/$name.txt 2>&1 || { echo "FAILED: $name" | tee -a $OUT/summary.txt; tail -n 5 $OUT/$name.txt; exit 1; }
  tail -n 1 $OUT/$name.txt | cut -c 1-300 | tee -a $OUT/summary.txt
}
case "${PART:-1}" in
1)
  SOAK_STREAM=1 SOAK_POOLS=120 SOAK_SEED=301 run dummy_soak_stream 500 python tools/dummy_soak.py
  SOAK_STREAM=1 SOAK_MIXED=1 SOAK_POOLS=100 SOAK_SEED=302 run dummy_soak_mixed 500 python tools/dummy_soak.py
  FUZZ_ROUNDS=60 FUZZ_SEED=303 run dummy_abi_fuzz 300 python tools/dummy_abi_fuzz.py
  SOAK_SECONDS=90 SOAK_SEED=308 run sharded_soak 240 python tools/sharded_soak.py
  ;;
2)
  SOAK_RECORDS=3000000 SOAK_SEED=304 run ygo_soak 700 python tools/ygo_soak.py
  FUZZ_ROUNDS=30 FUZZ_SEED=305 run ygo_record_fuzz 300 python tools/ygo_record_fuzz.py
  FUZZ_ROUNDS=10 FUZZ_SEED=306 run ygo_abi_fuzz 300 python tools/ygo_abi_fuzz.py
  ;;
3)
  SOAK_ROWS=2500000 SOAK_SEED=307 run ygo_pool_soak 800 python tools/ygo_pool_soak.py
  LONG_ROLLOUTS=600000 LONG_SERVER_STEPS=2000000 run dummy_long_run 350 python tools/dummy_long_run.py
  ;;
esac
echo "part ${PART:-1} done" | tee -a $OUT/summary.txt
