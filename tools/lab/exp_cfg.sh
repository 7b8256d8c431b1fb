#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
python3 bench.py --config 4 --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','fit_s','predict_s','layer_fit_ms')})"
python3 bench.py --config 3 --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','fit_s','predict_s','layer_fit_ms')})"
