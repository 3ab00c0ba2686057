#!/bin/bash
# Product-library variants for A/B runs of compile-time options (tools/lib_variant_ab.py): only the translation units named in FILES
# (default "gemm attention") are recompiled, the other objects come from slime_amd/csrc/build (run `make product` first).
# usage: [FILES="api"] build_variants.sh name:"-Dflag=0 ..." ...
set -e
cd "$(dirname "$0")/../slime_amd/csrc"
ROOT=$(cd ../.. && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/slime_amd/csrc -Wno-unused-result"
mkdir -p "$ROOT/slime_amd/variants"
pids=()
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  d=build_var_$name; mkdir -p $d
  ( objs=""
    for f in gemm attention prefill rowwise router slicer patch_embed calib api; do
      if [[ " ${FILES:-gemm attention} " == *" $f "* ]]; then
        extra=""; [[ $f == attention || $f == prefill ]] && extra="-fno-honor-nans"
        /opt/rocm/bin/hipcc $FLAGS $defs $extra -c $f.hip -o $d/$f.o || exit 1
        objs="$objs $d/$f.o"
      else objs="$objs build/$f.o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $objs -o "$ROOT/slime_amd/variants/libslime_hip_$name.so" &&
    echo "built $name" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
