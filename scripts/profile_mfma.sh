#!/bin/bash
# usage: scripts/profile_mfma.sh <tag> [bench args]   (GPU box, repo root): matrix-core counters of the bench kernels
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-mica --no-host $@"
i=20
for PMC in "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 bench.py $ARGS > $OUT/pmc$i.log 2>&1 || echo "pass $i failed: $PMC"
done
python3 scripts/summarize_prof.py $OUT > $OUT/summary.json
cat $OUT/summary.json
