# same box: the null's simulator gathering its tables from L2 (CMX_SIM_GATHER=1) vs tables of the current node in LDS
W=${1:-target}
for F in 1 0 1 0; do CMX_SIM_GATHER=$F timeout -k 10 300 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-mica --no-host 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gather=$F step_ms', round(d['ms_per_step'],2), 'map_launch_ms', round(d['roofline']['launch_ms'],2), 'value %.4g' % d['value'])"; done
