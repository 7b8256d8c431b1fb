#!/bin/bash
# one round of checks after a schedule change: the GPU tests, the bench line, the layer timings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/${1:-x}_tests.log 2>&1; tail -3 gpurun_out/${1:-x}_tests.log
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(value=round(d['value'],2), ms=round(d['ms_per_step'],3), stage=d['stage_ms'], roof=round(d['roofline']['frac'],3), chol=d.get('cholesky_frac_of_peak'))))"
python3 tools/layer_time.py 128 2048 5 2>/dev/null | tail -1
python3 tools/layer_time.py 64 4096 3 2>/dev/null | tail -1
