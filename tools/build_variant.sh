#!/bin/bash
# Builds libfdtd_amd.so with extra compiler flags into build/variants/<name>/ (for A/B runs on the GPU box).
# Usage: build_variant.sh <name> <flags...>
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/build/variant_obj/$name; out=$root/build/variants/$name
mkdir -p $obj $out
cd $root/fdtd_amd/csrc
for f in sweep2d sweep2d_tb sweep2d_tile sweep3d kernels context host_setup comm; do
  src=$f.hip; [ -f $src ] || src=$f.cpp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-parameter "$@" -I$root/include -I. -c -o $obj/$f.o $src 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libfdtd_amd.so $obj/*.o -ldl -lpthread
ls -la $out/libfdtd_amd.so
