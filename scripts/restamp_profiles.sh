#!/bin/bash
# GPU box, repo root: kernel traces + PMC passes of the three mapping workloads and the Mica legs, the traffic file stamped
# with the sha of the device sources, and the final bench lines -> gpurun_out/ (copy into profiles/ with
# scripts/collect_profiles.sh <round tag> afterwards, on the machine that holds the git tree)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r03}
scripts/profile_r03.sh ${R}_target --workload target > gpurun_out/prof_${R}_target.log 2>&1
scripts/profile_r03.sh ${R}_cfg3 --workload cfg3 > gpurun_out/prof_${R}_cfg3.log 2>&1
scripts/profile_r03.sh ${R}_cfg4 --workload cfg4 > gpurun_out/prof_${R}_cfg4.log 2>&1
scripts/profile_mica.sh ${R}_mica > gpurun_out/prof_${R}_mica.log 2>&1
scripts/profile_mica.sh ${R}_mica_gaps --gap-columns 1.0 > gpurun_out/prof_${R}_mica_gaps.log 2>&1
scripts/profile_mica.sh ${R}_mica_dna --alpha 4 > gpurun_out/prof_${R}_mica_dna.log 2>&1
python3 scripts/make_traffic_json.py ${R} target=${R}_target cfg3=${R}_cfg3 cfg4=${R}_cfg4 mica_cfg5=${R}_mica > gpurun_out/traffic.log 2>&1
cp profiles/traffic_${R}.json gpurun_out/traffic_${R}.json
for t in target cfg3 cfg4 mica mica_gaps mica_dna; do head -1 gpurun_out/prof_${R}_$t.log; done
python bench.py > gpurun_out/${R}_final_target_bench.json 2>/dev/null
python bench.py --workload cfg3 --no-mica > gpurun_out/${R}_final_cfg3_bench.json 2>/dev/null
python bench.py --workload cfg4 --no-mica > gpurun_out/${R}_final_cfg4_bench.json 2>/dev/null
python bench.py --workload cfg2 --steps 20 --warmup 3 --no-mica > gpurun_out/${R}_final_cfg2_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_cfg2/trace -- python3 bench.py --workload cfg2 --steps 20 --warmup 3 --no-cpu-baseline --no-host --no-mica > gpurun_out/prof_${R}_cfg2.log 2>&1
find gpurun_out/prof_${R}_cfg2/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_${R}_cfg2/kernel_stats.csv \;
python3 - <<PY
import json
for t in ("target", "cfg3", "cfg4", "cfg2"):
    d = json.loads([l for l in open("gpurun_out/${R}_final_%s_bench.json" % t) if l.startswith("{")][-1])
    r = d["roofline"]
    print(t, "value %.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], "launch %.2f" % r["launch_ms"], "frac %.3f" % r["frac"],
          "sim %.2f" % r["simulate_ms"], "rest %.2f" % r["rest_of_step_ms"], "traffic", r["traffic"], r.get("hbm_frac"))
    if "mica_cfg5" in d:
        m = d["mica_cfg5"]
        print("  mica", "%.4g" % m["value"], m["ms_per_step"], m["roofline"]["frac"], m["roofline"]["traffic"])
PY
