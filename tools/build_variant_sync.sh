#!/bin/bash
# usage: tools/build_variant_sync.sh NAME "-DFOO=1 ..."  -> variants/libdabhip_NAME.so (k_sync object rebuilt with the flags)
set -euo pipefail
cd "$(dirname "$0")/../dabtools_amd/csrc"
NAME=$1; FLAGS=$2
OUT=../../variants; mkdir -p $OUT/obj_$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $FLAGS -c k_sync.hip -o $OUT/obj_$NAME/k_sync.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $OUT/libdabhip_$NAME.so $OUT/obj_$NAME/k_sync.o $(ls build/*.o | grep -v '/k_sync.o$')
echo built $OUT/libdabhip_$NAME.so
