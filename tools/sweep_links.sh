#!/bin/bash
# A/B sweeps of the factorisation's schedule switches (cimrgp_amd/csrc/potrf.hip: struct Tuning), run on
# the GPU box from the repo root:   bash tools/sweep_links.sh chain | thresholds | rows
# Every setting runs in a process of its own (the switches are read once per process).
pt() { timeout -k 10 200 python tools/potrf_time.py $1 5 2>/dev/null; }
bn() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('posteriors/s', round(d['value'], 2), 'potrf_with_carried_rows', round(d['stage_ms']['potrf_with_carried_rows'], 3), 'potrf_alone', round(d['stage_ms']['potrf_alone'], 3))"; }
case "${1:-chain}" in
chain)        # the three forms of the panel chain against the default (by context)
    for c in default split wide quad; do
        echo "chain=$c"; if [ $c = default ]; then unset CIMRGP_CHAIN; else export CIMRGP_CHAIN=$c; fi
        for n in 2048 8192 16384; do pt $n; done; bn
    done ;;
thresholds)   # single-queue tail, head-first ordering, far pairing
    for tb in 2816 3840 4864 5888; do echo "tail_below=$tb"; CIMRGP_TAIL_BELOW=$tb pt 8192; CIMRGP_TAIL_BELOW=$tb pt 6144; done
    for hf in 0 4608 1073741824; do echo "head_first_above=$hf"; CIMRGP_HEAD_FIRST=$hf pt 8192; CIMRGP_HEAD_FIRST=$hf pt 16384; done
    for fp in 2048 4096 6144 8192; do echo "far_pair_above=$fp"; CIMRGP_FAR_PAIR=$fp pt 8192; CIMRGP_FAR_PAIR=$fp pt 12288; CIMRGP_FAR_PAIR=$fp bn; done ;;
rows)         # carried rows: start point, one queue against two
    for rs in 3584 4608 5632 6656 8192; do echo "rows_start_below=$rs"; CIMRGP_ROWS_START=$rs bn; done
    echo "rows on one queue"; CIMRGP_ROWS_ONE_QUEUE=1 bn ;;
esac
