# usage (GPU box, repo root): ab_mica_libs_taxa.sh TAXA libA.so libB.so ...   same box, alternating: Mica, 5 000 x 5 000 columns,
# TAXA taxa, without unknowns and with unknowns in every column
T=$1; shift
for R in 1 2; do for L in "$@"; do
  for G in 0.0 1.0; do
    COMAP_MI355X_LIB=$PWD/$L timeout -k 10 200 python scripts/bench_mica.py --taxa $T --steps 10 --gap-columns $G 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L taxa=$T gaps=$G', round(d['ms'],3), d['max_identity_residual'])"
  done
done; done
