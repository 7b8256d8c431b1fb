#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "block_posterior or lookahead or rows" 2>&1 | tail -3
for i in 1 2 3; do
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(value=round(d['value'],2), ms=round(d['ms_per_step'],3), stage=d['stage_ms'], roof=round(d['roofline']['frac'],3), chol=round(d['cholesky_frac_of_peak'],4))))"
done
python3 bench.py --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), d.get('parity_ok'), d.get('parity_rel_err_mean'), d.get('parity_rel_err_var_elementwise'))"
