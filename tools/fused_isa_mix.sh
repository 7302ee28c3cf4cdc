#!/bin/bash
# The static instruction mix of the default OFDM stage's symbol loop (ofdm_demap_kernel<false>, guarded build of k_fused.hip), priced in issue
# cycles: profiles/r06_fused_isa_mix.{txt,json} (bench.py reads the JSON; tests/test_bench_launch.py holds its source hash against the tree).
# CPU only: hipcc cross-compiles.  Re-run after every change of k_fused.hip / fft_core.hpp / device_types.hpp -- the block labels move.
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S="$R/dabtools_amd/csrc"
A=$(mktemp /tmp/fused_XXXX.s)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$S/k_fused.hip" -o "$A" 2>/dev/null
SHA=$(cat "$S/k_fused.hip" "$S/fft_core.hpp" "$S/device_types.hpp" | sha256sum | cut -d' ' -f1)
# The loop body holds two symbols.  Weights (read off the listing this script prints; the block structure is the same for both symbols):
#   0    the per-sample path of frames whose window leaves what the call read (stale tail, sdr_fifo.c:56-59): symbol 75 of a frame after a negative shift only
#   0.5  the byte pack of the previous symbol's decisions by 96 of the 256 threads: two of the four waves enter
#   R    the parity guard's per-bin repeat (entered by a wave when one of its lanes trips the per-thread threshold) and its list append:
#        not priced by weight -- their mix prices the instructions the measured count per transform has beyond the weighted static count
# (round 6: the list append now also holds the proven level's per-bin test -- BB0_56 / BB0_86, 103 instructions, entered once per listed candidate: weight 0 like the append before it;
#  the second symbol's blocks moved up by three labels)
W="${FUSED_WEIGHTS:-BB0_27=0,BB0_29=0,BB0_67=0,BB0_69=0,BB0_41=0.5,BB0_43=0.5,BB0_45=0.5,BB0_73=0.5,BB0_75=0.5,BB0_48=0,BB0_50=0,BB0_52=0,BB0_56=0,BB0_78=0,BB0_80=0,BB0_82=0,BB0_86=0,BB0_40=0,BB0_64=0}"
RM="${FUSED_REMAINDER:-BB0_48,BB0_50,BB0_52,BB0_78,BB0_80,BB0_82}"
python3 "$R/tools/isa_mix.py" --asm "$A" --kernel ofdm_demap_kernelILb0 --per-iteration 0.5 --weights "$W" --remainder "$RM" --source-sha "$SHA" > "$R/profiles/r06_fused_isa_mix.txt"
tail -n 1 "$R/profiles/r06_fused_isa_mix.txt" > "$R/profiles/r06_fused_isa_mix.json"
rm -f "$A"
cat "$R/profiles/r06_fused_isa_mix.txt" | cut -c1-200
