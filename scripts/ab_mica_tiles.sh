# same box, same library: packed (3 columns per 64-row block) vs one-column-per-tile Mica kernel, without and with gaps
for M in 1 0 1 0; do
  CMX_MICA_TILES=$M timeout -k 10 200 python scripts/bench_mica.py --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one_column_tiles=$M', round(d['ms'],3), d['max_identity_residual'])"
  CMX_MICA_TILES=$M timeout -k 10 200 python scripts/bench_mica.py --steps 5 --gap-columns 1.0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one_column_tiles=$M gaps', round(d['ms'],3), d['max_identity_residual'])"
done
