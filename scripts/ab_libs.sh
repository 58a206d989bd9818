# usage: ab_libs.sh workload lib1 lib2 ...   (same box, alternating)
W=$1; shift
for i in 1 2; do for L in "$@"; do COMAP_MI355X_LIB=$PWD/$L timeout -k 10 300 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-mica --no-host 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['roofline']['launch_ms'],3), round(d['roofline']['frac'],4))"; done; done
