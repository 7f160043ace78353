#!/bin/bash
# builds nlzm_amd/libnlzm_hip_prof.so (-DNLZM_PROFILE: per-section cycle accounting) next to the product library
# (the file is listed in .gpurunignore: build it on the GPU box inside the gpurun command, or under another name -- round 6 used
#  nlzm_amd/libnlzm_exp_prof.so -- and load it through NLZM_LIB; the profile build also counts what a hot bin's wave spends its steps on, by section)
set -e
cd "$(dirname "$0")/../nlzm_amd/csrc"
mkdir -p /tmp/nlzm_prof_build
/opt/rocm/bin/hipcc -DNLZM_PROFILE $EXTRA -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -c -o /tmp/nlzm_prof_build/k.o nlzm_kernels.hip &
/opt/rocm/bin/hipcc -DNLZM_PROFILE $EXTRA -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-undefined-inline -x hip -c -o /tmp/nlzm_prof_build/h.o nlzm_hip.cpp &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libnlzm_hip_prof.so /tmp/nlzm_prof_build/k.o /tmp/nlzm_prof_build/h.o
