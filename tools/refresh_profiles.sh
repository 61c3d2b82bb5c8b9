#!/bin/bash
# Everything profiles/<round>/ holds, re-made on the GPU box (from the repo root), in parts that each fit one gpurun call:
#   tools/refresh_profiles.sh 1   default bench line + kernel trace / PMC passes of the three 4096^2 workloads
#   tools/refresh_profiles.sh 2   the 8192^2 workloads + the plain bench lines of every workload
#   tools/refresh_profiles.sh 3   K2/K3 micro-benches, ICP search, SQ counters of the chain kernel (tools/prof_round_micro.sh)
# Results land in gpurun_out/<round>/; copy what is to be judged into profiles/<round>/.
R=${QS_PROF_ROUND:-r03}
mkdir -p gpurun_out/$R
case "$1" in
1)
  python3 bench.py > gpurun_out/$R/bench_c1_default_run.json 2> gpurun_out/$R/bench_c1_default_run.err; echo "default rc=$?"
  bash tools/prof_round.sh c1_4096 && echo c1_4096 ok
  bash tools/prof_round.sh c3_4096 --workload c3 && echo c3_4096 ok
  bash tools/prof_round.sh adv_4096 --workload adv && echo adv_4096 ok
  ;;
2)
  bash tools/prof_round.sh c1_8192 --grid 8192 && echo c1_8192 ok
  bash tools/prof_round.sh c3_8192 --workload c3 --grid 8192 && echo c3_8192 ok
  bash tools/prof_round.sh adv_8192 --workload adv --grid 8192 && echo adv_8192 ok
  for wl in c3 adv; do python3 bench.py --workload $wl --no-micro > gpurun_out/$R/bench_$wl.json 2> gpurun_out/$R/bench_$wl.err; echo "$wl rc=$?"; done
  for wl in c1 c3 adv; do python3 bench.py --workload $wl --grid 8192 --no-micro > gpurun_out/$R/bench_${wl}_8192.json 2> gpurun_out/$R/bench_${wl}_8192.err; echo "$wl 8192 rc=$?"; done
  ;;
3)
  bash tools/prof_round_micro.sh && echo micro ok
  ;;
esac
