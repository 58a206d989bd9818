#!/bin/bash
# after scripts/restamp_profiles.sh came back through gpurun: gpurun_out/ -> profiles/ (tracked)
R=${1:-r03}
cp gpurun_out/traffic_${R}.json profiles/traffic_${R}.json
for t in target cfg3 cfg4; do
  cp gpurun_out/prof_${R}_$t/summary.json profiles/${R}_${t}_pmc_summary.json
  cp gpurun_out/prof_${R}_$t/kernel_stats.csv profiles/${R}_${t}_kernel_stats.csv
  grep '^{' gpurun_out/${R}_final_${t}_bench.json | tail -1 > profiles/${R}_final_${t}_bench.json
done
cp gpurun_out/prof_${R}_cfg2/kernel_stats.csv profiles/${R}_cfg2_kernel_stats.csv
grep '^{' gpurun_out/${R}_final_cfg2_bench.json | tail -1 > profiles/${R}_final_cfg2_bench.json
cp gpurun_out/prof_${R}_mica/summary.json profiles/${R}_mica_cfg5_pmc_summary.json
cp gpurun_out/prof_${R}_mica/kernel_stats.csv profiles/${R}_mica_cfg5_kernel_stats.csv
cp gpurun_out/prof_${R}_mica_gaps/summary.json profiles/${R}_mica_cfg5_gaps_pmc_summary.json
cp gpurun_out/prof_${R}_mica_gaps/kernel_stats.csv profiles/${R}_mica_cfg5_gaps_kernel_stats.csv
cp gpurun_out/prof_${R}_mica_dna/summary.json profiles/${R}_mica_cfg5_dna_pmc_summary.json
cp gpurun_out/prof_${R}_mica_dna/kernel_stats.csv profiles/${R}_mica_cfg5_dna_kernel_stats.csv
python3 -c "import bench, json; print('sha', bench.kernel_source_sha(), json.load(open('profiles/traffic_${R}.json'))['kernel_source_sha'])"
