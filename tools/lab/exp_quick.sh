#!/bin/bash
# quick check after a kernel change: look-ahead parity tests, stand-alone update, potrf sweep, bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "lookahead or rows or potrf or gemm or syrk or one_call" 2>&1 | tail -2
python3 tools/gemm_bench.py --m 7936,5888 --k 256 --reps 20 --check 2>/dev/null
python3 tools/potrf_sweep.py --sizes 4096,8192,16384 2>/dev/null
python3 tools/potrf_sweep.py --sizes 8192 --rows 2>/dev/null
for i in 1 2; do
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(value=round(d['value'],2), ms=round(d['ms_per_step'],3), potrf_rows=round(d['stage_ms']['potrf_with_carried_rows'],3), potrf_alone=round(d['stage_ms']['potrf_alone'],3), roof=round(d['roofline']['frac'],3), chol=round(d.get('cholesky_frac_of_peak'),4))))"
done
