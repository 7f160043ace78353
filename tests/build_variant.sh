#!/bin/bash
# builds a variant of the library for A/B runs on one box:  tests/build_variant.sh NAME -DFLAG...   -> nlzm_amd/libnlzm_hip_NAME.so
set -e
name=$1; shift
cd "$(dirname "$0")/../nlzm_amd/csrc"
d=/tmp/nlzm_var_$name; mkdir -p $d
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -c -o $d/k.o nlzm_kernels.hip &
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-undefined-inline -x hip -c -o $d/h.o nlzm_hip.cpp &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libnlzm_hip_$name.so $d/k.o $d/h.o
