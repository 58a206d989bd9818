# same box: the eight-wave kernel for every tile (CMX_MICA_TILES=3) against the four-wave kernel (grid sizes), with and
# without unknowns, three rounds
for R in 1 2 3; do
  CMX_MICA_TILES=3 timeout -k 10 200 python scripts/bench_mica.py --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eight-wave', round(d['ms'],3))"
  for G in 512 1024 2048; do
    CMX_MICA4_GRID=$G timeout -k 10 200 python scripts/bench_mica.py --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('four-wave grid=$G', round(d['ms'],3), d['max_identity_residual'])"
  done
  CMX_MICA_TILES=3 timeout -k 10 200 python scripts/bench_mica.py --steps 5 --gap-columns 1.0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('eight-wave gaps', round(d['ms'],3))"
  timeout -k 10 200 python scripts/bench_mica.py --steps 5 --gap-columns 1.0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('four-wave, gaps', round(d['ms'],3), d['max_identity_residual'])"
  timeout -k 10 200 python scripts/bench_mica.py --steps 5 --gap-columns 0.1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('four-wave, gaps 10%', round(d['ms'],3), d['max_identity_residual'])"
done
