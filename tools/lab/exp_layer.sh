#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python3 tools/layer_time.py 128 2048 3 2>/dev/null | tail -1
python3 tools/layer_time.py 64 4096 3 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lay_stats -- python3 tools/layer_time.py 128 2048 2 > gpurun_out/lay.log 2>&1
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob("gpurun_out/lay_stats/*/*kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%9.1f us avg" % (float(r["AverageNs"])/1e3), "%6.1f %%" % float(r["Percentage"]))
PY
rm -rf gpurun_out/lay_stats
