#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for mode in "" "--nccl-world1" "--nccl-world1 --rows-queues 1"; do
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline $mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],3), d['config'].get('backend'), d['config'].get('rows_queues'), d['stage_ms']['reduce'])"
done
CIMRGP_BENCH_REHEARSAL=gloo python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
python3 bench.py --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), d.get('parity_ok'), d.get('parity_rel_err_mean'), d.get('parity_rel_err_var_elementwise'))"
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py -x -q -k "rccl or bench" 2>&1 | tail -2
