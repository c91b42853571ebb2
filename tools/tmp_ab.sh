python -m pytest tests/test_gpu_switches.py tests/test_gpu_corners.py -x -q -k "fused or pipeline or chunked" 2>&1 | tail -3
run() { echo -n "$1 $2: "; env $1 python bench.py --no-cpu-baseline --no-extra $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['launch_ms'])"; }
for i in 1 2; do
run "X=1" ""
run "TROYN_TAIL_ORDER=poly" ""
done
