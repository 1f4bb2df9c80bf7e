#!/bin/bash
# kernel trace of the config-2 bench step (gaps, no-op slots): bash tools/trace_c2.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-trace}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=${DEBUG_CLR_GRAPH_PACKET_CAPTURE:-0} ROC_AQL_QUEUE_SIZE=524288
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --batch 0 --no-configs --no-inexact > $O/bench.json 2> $O/bench.err
echo "rc=$?"
python3 $R/tools/trace_gaps.py $O/t > $O/gaps.txt 2>&1
cat $O/gaps.txt
cp $(find $O/t -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/t
