# usage: ab_step.sh workload lib1 lib2 ...   (same box, alternating): whole step, mapping launch
W=$1; shift
for i in 1 2; do for L in "$@"; do COMAP_MI355X_LIB=$PWD/$L timeout -k 10 300 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-mica --no-host 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L step_ms', round(d['ms_per_step'],2), 'launch_ms', round(d['roofline']['launch_ms'],2), 'value %.4g' % d['value'])"; done; done
