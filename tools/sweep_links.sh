for mode in 0 1 2; do for n in 8192 16384 32768; do echo "links=$mode n=$n"; CIMRGP_LINKS=$mode timeout -k 10 200 python tools/potrf_time.py $n 4 2>/dev/null; done; done
for tb in 4864 5632 6400 7168 9000; do echo "tail_below=$tb"; CIMRGP_TAIL_BELOW=$tb timeout -k 10 100 python tools/potrf_time.py 8192 5 2>/dev/null; done
for mode in 0 1 2; do echo "bench links=$mode"; CIMRGP_LINKS=$mode timeout -k 10 300 python bench.py --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stage_ms'])"; done
