# A/B sweeps of the chain schedule (run on the GPU box): bash tools/sweep_links.sh
pt() { timeout -k 10 200 python tools/potrf_time.py $1 5 2>/dev/null; }
bn() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms']['potrf_with_carried_rows'], d['stage_ms']['potrf_alone'])"; }
for tb in 2816 3840 4864; do echo "tail_below=$tb"; CIMRGP_TAIL_BELOW=$tb pt 8192; CIMRGP_TAIL_BELOW=$tb pt 6144; done
for c in hybrid quad; do echo "chain=$c"; if [ $c = hybrid ]; then unset CIMRGP_CHAIN; else export CIMRGP_CHAIN=$c; fi; bn; bn; done
