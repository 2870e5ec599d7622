#!/bin/bash
# collect PMC counters for one configuration: bash tools/pmc.sh "<run_one args>" "<counter list 1>" "<counter list 2>" ...
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
args="$1"; shift
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc$i -o p -- python "$GRAFT_REPO_ROOT/tools/run_one.py" $args > /tmp/pmc$i.log 2>&1)
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  t=$(find /tmp/pmc$i -name "*kernel_trace.csv" | head -1)
  python - "$f" "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'fat5' not in r['Kernel_Name']: continue
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
if len(sys.argv) > 2 and sys.argv[2]:  # the launches' own durations in this (counter-collecting) run: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration
    for r in csv.DictReader(open(sys.argv[2])):
        if 'fat5' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]]['duration_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:32s} avg {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
done
