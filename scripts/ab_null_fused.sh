# same box, same library: the null with the simulator inside the mapping waves (CMX_NULL_FUSED=1, round 1's arrangement) vs
# simulate first at full occupancy, then map (default).  usage: ab_null_fused.sh workload
W=${1:-target}
for F in 1 0 1 0; do CMX_NULL_FUSED=$F timeout -k 10 300 python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-mica --no-host 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$F step_ms', round(d['ms_per_step'],2), 'map_launch_ms', round(d['roofline']['launch_ms'],2), 'frac', round(d['roofline']['frac'],4), 'value %.4g' % d['value'])"; done
