#!/bin/bash
# On the GPU box: time the null-mode mapping kernel for each ablation build (scripts/build_ablations.sh).
for n in "$@"; do
  COMAP_MI355X_LIB=$PWD/build/abl/libcmx_abl$n.so timeout -k 10 200 python bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-mica --no-host  \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate $n', 'ms_per_step', round(d['ms_per_step'],2), 'null_launch_ms', round(d['roofline']['launch_ms'],2))" || exit 1
done
