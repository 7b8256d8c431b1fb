# A/B sweeps of the chain schedule (run on the GPU box): bash tools/sweep_links.sh
pt() { timeout -k 10 200 python tools/potrf_time.py $1 5 2>/dev/null; }
bn() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms']['potrf_with_carried_rows'], d['stage_ms']['potrf_alone'])"; }
for fp in 8192 6144 4096 2048; do echo "far_pair_above=$fp"; CIMRGP_FAR_PAIR=$fp pt 8192; CIMRGP_FAR_PAIR=$fp pt 12288; CIMRGP_FAR_PAIR=$fp bn; done
