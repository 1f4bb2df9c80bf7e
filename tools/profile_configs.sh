#!/bin/bash
# rocprofv3 passes over the config-3 (Lasso) and config-5 (portfolio) probes: --kernel-trace --stats, --pmc FETCH_SIZE,
# --pmc WRITE_SIZE (separate runs, as the pool requires), condensed by tools/profile_summary.py.
#   usage (on the GPU box): bash tools/profile_configs.sh <tag>     -> gpurun_out/<tag>/config{3,5}.json
# DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: with ROCm 7.2's graph packet capture on, rocprofiler-sdk's queue interceptor reads a
# captured packet batch past the end of the 16384-packet AQL ring once the ring wraps (host SIGSEGV in
# librocprofiler-sdk.so under hipGraphLaunch; DESIGN.md "Profiling").  Kernel durations and counters are unaffected.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-prof}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
# the same interceptor fault also shows without packet capture once a long graph's packets straddle the ring end (config 3,
# --pmc pass): a ring large enough for the whole probe never wraps
export ROC_AQL_QUEUE_SIZE=${ROC_AQL_QUEUE_SIZE:-524288}
run() {
  local name=$1; shift
  timeout -k 10 900 "$@" > $O/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $O/rc.txt
  if [ $rc -ne 0 ]; then tail -5 $O/$name.log; echo "stopping" | tee -a $O/rc.txt; exit 1; fi
}
for cfg in ${CONFIGS:-portfolio lasso}; do
  if [ $cfg = lasso ]; then export PROBE_MAX_ITER=${LASSO_MAX_ITER:-200}; id=config3; else unset PROBE_MAX_ITER; id=config5; fi
  run ${id}_trace rocprofv3 --kernel-trace --stats --output-format csv -d $O/${id}_trace -- python3 $R/tools/c5probe.py $cfg
  run ${id}_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${id}_fetch -- python3 $R/tools/c5probe.py $cfg
  run ${id}_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${id}_write -- python3 $R/tools/c5probe.py $cfg
  python3 $R/tools/profile_summary.py $O/${id}_trace $O/${id}_fetch $O/${id}_write $O/${id}.json > $O/${id}_summary.txt 2>&1
  cp $(find $O/${id}_trace -name "*kernel_stats.csv" | head -1) $O/${id}_kernel_stats.csv
  grep -v "^\[osqp\|^W2\|^E2" $O/${id}_trace.log | tail -8 > $O/${id}_probe.txt
  rm -rf $O/${id}_trace $O/${id}_fetch $O/${id}_write
done
cat $O/rc.txt; du -sh $O
