#!/bin/bash
# Diagnostic builds of the library with parts of the mapping kernel removed: timing only, results are wrong.
# The switches (CMX_ABLATE, CMX_TIMING, CMX_FUSED_SIM, CMX_PROBE, CMX_MICA_ABLATE, HC_ABLATE) are NOT in the product source
# any more: this script builds from the tagged snapshot that still carries them (README.md in this directory).
# Usage: scripts/experiments/ablations/build_ablations.sh "1 2 3" -> build/abl/libcmx_abl<N>.so
set -e
ROOT="$(cd "$(dirname "$0")/../../.." && pwd)"
TAG=${CMX_ABLATION_TAG:-ablation-hooks-r03}
rm -rf "$ROOT/build/abl/src" && mkdir -p "$ROOT/build/abl/src/comap_amd" "$ROOT/build/abl/src/include"
git -C "$ROOT" archive "$TAG" comap_amd/csrc include | tar -x -C "$ROOT/build/abl/src"
cd "$ROOT/build/abl/src/comap_amd/csrc"
mkdir -p ../../../../abl && ln -sfn "$ROOT/build/abl" ../../build_abl_out 2>/dev/null || true
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1"
for n in ${1:-1 2 3 4 5 6}; do
  if [ "${n:0:1}" = "m" ]; then
    /opt/rocm/bin/hipcc $FLAGS -DCMX_MICA_ABLATE=${n:1} -shared cmx_kernels.hip cmx_mica4.hip cmx_stat_mi.hip cmx_cluster.hip cmx_mica_post.hip cmx_variants.hip -x hip cmx_api.cpp cmx_host_model.cpp -o $ROOT/build/abl/libcmx_mica_abl${n:1}.so &
  elif [ "$n" = "t" ]; then   # phase timing with s_memtime, printed by waves 0 and 777 of the null kernel
    /opt/rocm/bin/hipcc $FLAGS -DCMX_TIMING -shared cmx_kernels.hip cmx_mica4.hip cmx_stat_mi.hip cmx_cluster.hip cmx_mica_post.hip cmx_variants.hip -x hip cmx_api.cpp cmx_host_model.cpp -o $ROOT/build/abl/libcmx_timing.so &
  else
    /opt/rocm/bin/hipcc $FLAGS -DCMX_ABLATE=$n -shared cmx_kernels.hip cmx_mica4.hip cmx_stat_mi.hip cmx_cluster.hip cmx_mica_post.hip cmx_variants.hip -x hip cmx_api.cpp cmx_host_model.cpp -o $ROOT/build/abl/libcmx_abl$n.so &
  fi
done
wait
# Mica packed kernel (timing only): scripts/build_ablations.sh "m1 m2 m3 m4" -> -DCMX_MICA_ABLATE=1 no table lookups,
# 2 no matrix products, 3 no one-hot expansion, 4 no epilogue sums / stores (build/abl/libcmx_mica_abl<N>.so)
ls -la $ROOT/build/abl
