#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out; TAG=${1:-x}
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_layer.py -x -q -m gpu > $OUT/${TAG}_tests.log 2>&1 || { tail -40 $OUT/${TAG}_tests.log; echo TESTS FAILED; exit 1; }
tail -2 $OUT/${TAG}_tests.log
for n in 1024 2048 4096 6144 8192 16384; do timeout -k 10 100 python3 tools/potrf_time.py $n 4 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(value=round(d['value'],2), ms=round(d['ms_per_step'],3), stage=d['stage_ms'], chol_frac=round(d['cholesky_frac_of_peak'],3), roof=round(d['roofline']['frac'],3))))"
