#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   -> winterfell_amd/variants/<name>/libwinterfell_hip.so
set -e
cd "$(dirname "$0")/../winterfell_amd/csrc"
name=$1; shift
out=../variants/$name
mkdir -p $out/obj
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c $f -o $out/obj/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libwinterfell_hip.so $out/obj/*.o
rm -rf $out/obj
