# usage (GPU box, repo root): ab_step_libs.sh workload lib1 lib2 ...   same box, alternating: whole step, mapping launch, rest of the step
W=$1; shift
for i in 1 2; do for L in "$@"; do COMAP_MI355X_LIB=$PWD/$L timeout -k 10 300 python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-mica --no-host 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L $W step', round(d['ms_per_step'],3), 'launch', round(r['launch_ms'],3), 'rest', round(r['rest_of_step_ms'],3))"; done; done
