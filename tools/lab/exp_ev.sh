#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for f in "" "--no-launch-events" "" "--no-launch-events" "" "--no-launch-events"; do
python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline $f 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', json.dumps(dict(value=round(d['value'],2), ms=round(d['ms_per_step'],3), potrf_rows=round(d['stage_ms']['potrf_with_carried_rows'],3))))"
done
