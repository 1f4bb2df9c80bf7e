#!/bin/bash
# rocprofv3 passes over the headline bench (config 2 + the batch leg): --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE
# (separate runs), condensed by tools/profile_summary.py into gpurun_out/<tag>/config2.json.
#   usage (on the GPU box): bash tools/profile_bench.sh <tag>
# Environment: see tools/profile_configs.sh (work-arounds for the profiler's queue interceptor).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-profb}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 ROC_AQL_QUEUE_SIZE=${ROC_AQL_QUEUE_SIZE:-524288}
ARGS="--steps 4 --warmup 1 --no-cpu --no-configs --no-inexact"
run() {
  local name=$1; shift
  timeout -k 10 900 "$@" > $O/$name.json 2> $O/$name.err
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/rc.txt
  if [ $rc -ne 0 ]; then tail -5 $O/$name.err; exit 1; fi
}
run trace rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_trace -- python3 $R/bench.py $ARGS
run fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/t_fetch -- python3 $R/bench.py $ARGS
run write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/t_write -- python3 $R/bench.py $ARGS
python3 $R/tools/profile_summary.py $O/t_trace $O/t_fetch $O/t_write $O/config2.json > $O/summary.txt 2>&1
cp $(find $O/t_trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/t_trace $O/t_fetch $O/t_write
cat $O/rc.txt; head -30 $O/summary.txt
