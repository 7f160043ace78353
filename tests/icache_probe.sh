cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/icache
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -iE "ICACHE|IFETCH|INST_CACHE|SQC_" | head -30 > $OUT/list.txt
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $OUT/pmc -- python3 tests/gpu_probe.py 3 20 1 > $OUT/log.txt 2>&1
find $OUT -name "*counter_collection.csv" | head -3
F=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r.get("Kernel_Name","")[:40], r.get("Counter_Name"))] += float(r.get("Counter_Value") or 0)
for k, v in sorted(agg.items()): print(k, v)
PY
cat $OUT/list.txt | head -20; tail -3 $OUT/log.txt
