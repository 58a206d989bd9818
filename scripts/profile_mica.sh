#!/bin/bash
# usage (GPU box, repo root): scripts/profile_mica.sh <tag> [bench_mica args...]
# kernel trace + separate PMC passes of scripts/bench_mica.py (cfg 5), summarised like scripts/profile_r03.sh
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_mica.py $ARGS > $OUT/trace.log 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 scripts/bench_mica.py $ARGS > $OUT/pmc$i.log 2>&1 || echo "pass $i failed: $PMC"
done
python3 scripts/summarize_prof.py $OUT > $OUT/summary.json
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<PY
import json
d = json.load(open("$OUT/summary.json"))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1].get("avg_ms", 0) * kv[1].get("calls", 0))[:7]:
    print(k[:90], v.get("avg_ms"), v.get("calls"), v.get("vgpr"), v.get("scratch"))
    if "mica_mfma" in k:
        for c, x in sorted(v.get("pmc", {}).items()):
            print("   ", c, x)
PY
