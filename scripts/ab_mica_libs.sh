# usage (GPU box, repo root): ab_mica_libs.sh libA.so libB.so ...   same box, alternating: Mica cfg 5 without unknowns, with
# unknowns in every column, in a tenth of the columns
for R in 1 2; do for L in "$@"; do
  for G in 0.0 1.0 0.1; do
    COMAP_MI355X_LIB=$PWD/$L timeout -k 10 200 python scripts/bench_mica.py --steps 10 --gap-columns $G 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L gaps=$G', round(d['ms'],3), d['max_identity_residual'])"
  done
done; done
