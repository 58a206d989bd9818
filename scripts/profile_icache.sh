#!/bin/bash
# usage: scripts/profile_icache.sh <tag>   (GPU box, repo root): instruction/scalar-cache counters of the mapping kernel
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline $@"
i=10
for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ" \
           "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" ; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 bench.py $ARGS > $OUT/pmc$i.log 2>&1 || echo "pass $i failed: $PMC"
done
python3 scripts/summarize_prof.py $OUT > $OUT/summary.json
cat $OUT/summary.json
