#!/bin/bash
# round-5 GPU call 24: kernel averages of the skinny solves (potrs n = 8192, q = 2) and of the prediction tail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/potrs_tr -- python3 tools/potrs_time.py 8192 2 5 > gpurun_out/potrs_time.out 2> gpurun_out/potrs_tr.err || { tail -5 gpurun_out/potrs_tr.err; exit 1; }
tail -3 gpurun_out/potrs_time.out
f=$(find gpurun_out/potrs_tr -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|cimrgp::|void ", "", r["Name"])[:60]
    if "bwd" in name or "fwd" in name or "transpose" in name or "predict" in name:
        print("%-62s calls %5s total_ms %9.3f avg_us %9.2f min %s max %s" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["MinNs"], r["MaxNs"]))
PY
t=$(find gpurun_out/potrs_tr -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bwd" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-32:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%8.1f %8.1f %6.1f %s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][40:70], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
PY
rm -rf gpurun_out/potrs_tr
