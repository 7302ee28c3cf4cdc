#!/bin/bash
# variants/libdabhip_k1lb4.so: the in-tree library with the K1 chain kernel held to 128 VGPRs (__launch_bounds__(512, 4)), for tools/gpu/k1lb4.sh.
# Needs an up-to-date dabtools_amd/csrc/build/ (make -C dabtools_amd/csrc).
set -e
cd "$(dirname "$0")/../dabtools_amd/csrc"
sed 's/__global__ __launch_bounds__(kThreads) void sync_scan_kernel(/__global__ __launch_bounds__(kThreads, 4) void sync_scan_kernel(/' k_sync.hip > k_sync_lb4_tmp.hip
trap 'rm -f k_sync_lb4_tmp.hip' EXIT
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -c k_sync_lb4_tmp.hip -o ../../variants/k_sync_lb4.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "sync_scan_kernelILb1" | grep -E "VGPRs|Scratch|Occupancy"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../variants/libdabhip_k1lb4.so ../../variants/k_sync_lb4.o build/k_fft.o build/k_fused.o build/k_fused_plain.o build/k_fused_soft.o build/k_decode.o build/k_vitwave.o build/k_synth.o build/k_parity.o build/k_probe.o build/engine.o build/capi.o build/multi.o build/synth.o build/error.o
echo built variants/libdabhip_k1lb4.so
