#!/bin/bash
# Timings and counters of tools/mfma_dense_probe (MFMA vs VALU on the dense blocks of config 5): bash tools/profile_mfma.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-mfma}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
$R/tools/mfma_dense_probe 200 > $O/timings.txt 2>&1; echo "plain rc=$?" | tee $O/rc.txt
cat $O/timings.txt
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/p_$tag -- $R/tools/mfma_dense_probe 20 > $O/p_$tag.log 2>&1
  echo "$tag rc=$?" | tee -a $O/rc.txt
done
python3 - <<PY
import csv, glob, collections, json, statistics
out = collections.defaultdict(dict)
for f in glob.glob("$O/p_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = statistics.median(v)
json.dump(out, open("$O/counters.json", "w"), indent=1)
for k, v in sorted(out.items()):
    print(k, v)
PY
rm -rf $O/p_*/
cat $O/rc.txt
