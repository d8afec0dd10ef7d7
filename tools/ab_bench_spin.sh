for rep in 1 2 3 4; do for sp in 1 0; do
QS_BENCH_SPIN=$sp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-info-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spin=$sp', round(d['value']/1e6,2), round(d['ms_per_step']*1e3,2), round(d['roofline']['kernel_ms']*1e3,2))"
done; done
