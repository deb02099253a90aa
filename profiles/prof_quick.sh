cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pq; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o p -- python $R/bench.py --no_cpu_baseline --no_legs --no_planner --min_seconds 1 > /tmp/pq.json 2>/dev/null
python -c "
import json; r=json.load(open('/tmp/pq.json')); print(r['value'], r['ms_per_step'])"
f=$(find /tmp/pq -name "*kernel_stats.csv" | head -1)
STEPS=$(python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
print(max([int(r["Calls"]) for r in rows if "sample_group_kernel" in r["Name"]] or [1]))
PY
)
python $R/profiles/kernel_breakdown.py $f $STEPS | head -10
