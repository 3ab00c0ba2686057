#!/bin/bash
# Product-library variants for A/B runs of compile-time options (tools/lib_variant_ab.py): only gemm.hip / attention.hip are recompiled,
# the other objects come from slime_amd/csrc/build (run `make product` first).  usage: build_variants.sh name:"-Dflag=0 ..." ...
set -e
cd "$(dirname "$0")/../slime_amd/csrc"
ROOT=$(cd ../.. && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/slime_amd/csrc -Wno-unused-result"
mkdir -p "$ROOT/slime_amd/variants"
pids=()
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  d=build_var_$name; mkdir -p $d
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c gemm.hip -o $d/gemm.o &&
    /opt/rocm/bin/hipcc $FLAGS $defs -fno-honor-nans -c attention.hip -o $d/attention.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $d/gemm.o $d/attention.o \
        build/prefill.o build/rowwise.o build/router.o build/slicer.o build/api.o -o "$ROOT/slime_amd/variants/libslime_hip_$name.so" &&
    echo "built $name" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
