cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in build/abl/libcmx_prev.so comap_amd/libcomap_mi355x.so; do
  T=$(basename $L .so)
  COMAP_MI355X_LIB=$PWD/$L rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_lds_$T -- python3 bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-host --no-mica > gpurun_out/pmc_lds_$T.log 2>&1
  python3 - <<PY
import csv,glob
agg={}
for f in glob.glob("gpurun_out/pmc_lds_$T/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "map_kernel<20, 1, 1>" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print("$T", {k: sum(v)/len(v) for k,v in agg.items()})
PY
done
