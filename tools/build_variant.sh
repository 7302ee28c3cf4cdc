#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFOO=1 ..."  -> variants/libdabhip_NAME.so (k_fft / k_fused objects rebuilt with the flags)
set -e
cd "$(dirname "$0")/../dabtools_amd/csrc"
NAME=$1; FLAGS=$2
OUT=../../variants; mkdir -p $OUT/obj_$NAME
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $FLAGS"
$H -w -c k_fft.hip -o $OUT/obj_$NAME/k_fft.o &
$H -w -c k_fused.hip -o $OUT/obj_$NAME/k_fused.o &
$H -w -DDABHIP_FUSED_GUARD=0 -c k_fused.hip -o $OUT/obj_$NAME/k_fused_plain.o &
$H -w -DDABHIP_FUSED_SOFT=1 -c k_fused.hip -o $OUT/obj_$NAME/k_fused_soft.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libdabhip_$NAME.so $OUT/obj_$NAME/*.o $(ls build/*.o | grep -v -E '/(k_fft|k_fused|k_fused_plain|k_fused_soft)\.o$')
echo built $OUT/libdabhip_$NAME.so
