cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
bash tools/bench_variants.sh base lanerec2 lanerec4 base > $O/variants.txt 2>&1
cat $O/variants.txt
for V in lanerec2 lanerec4; do
DABHIP_LIB=$GRAFT_REPO_ROOT/variants/libdabhip_$V.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "viterbi or e2e or golden" 2>&1 | tail -n 2
done
