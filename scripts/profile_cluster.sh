#!/bin/bash
# usage: scripts/profile_cluster.sh <tag>   (run on the GPU box from the repo root)
# clustering null: kernel trace + separate PMC passes (the pool refuses --pmc combined with other trace domains)
TAG=$1
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--no-scipy --steps 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/bench_cluster.py $ARGS > $OUT/trace.log 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $GRAFT_REPO_ROOT/scripts/bench_cluster.py $ARGS > $OUT/pmc$i.log 2>&1 || echo "pass $i failed: $PMC"
done
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py $OUT > $OUT/summary.json
cat $OUT/summary.json | head -120
