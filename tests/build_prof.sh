#!/bin/bash
# builds nlzm_amd/libnlzm_hip_prof.so (-DNLZM_PROFILE: per-section cycle accounting) next to the product library
set -e
cd "$(dirname "$0")/../nlzm_amd/csrc"
mkdir -p /tmp/nlzm_prof_build
/opt/rocm/bin/hipcc -DNLZM_PROFILE $EXTRA -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -c -o /tmp/nlzm_prof_build/k.o nlzm_kernels.hip &
/opt/rocm/bin/hipcc -DNLZM_PROFILE $EXTRA -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-undefined-inline -x hip -c -o /tmp/nlzm_prof_build/h.o nlzm_hip.cpp &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libnlzm_hip_prof.so /tmp/nlzm_prof_build/k.o /tmp/nlzm_prof_build/h.o
