#!/bin/bash
# Diagnostic builds of the library with parts of the mapping kernel removed (CMX_ABLATE, see cmx_kernels.hip):
# timing only, results are wrong.  Usage: scripts/build_ablations.sh "1 2 3" -> build/abl/libcmx_abl<N>.so
set -e
cd "$(dirname "$0")/../comap_amd/csrc"
mkdir -p ../../build/abl
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1"
for n in ${1:-1 2 3 4 5 6}; do
  if [ "$n" = "t" ]; then   # phase timing with s_memtime, printed by waves 0 and 777 of the null kernel
    /opt/rocm/bin/hipcc $FLAGS -DCMX_TIMING -shared cmx_kernels.hip cmx_cluster.hip cmx_mica_post.hip -x hip cmx_api.cpp cmx_host_model.cpp -o ../../build/abl/libcmx_timing.so &
  else
    /opt/rocm/bin/hipcc $FLAGS -DCMX_ABLATE=$n -shared cmx_kernels.hip cmx_cluster.hip cmx_mica_post.hip -x hip cmx_api.cpp cmx_host_model.cpp -o ../../build/abl/libcmx_abl$n.so &
  fi
done
wait
ls -la ../../build/abl
