# Produces the round's profile artefacts under gpurun_out/profiles_new/ (copy the ones to keep into profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
rm -rf $O && mkdir -p $O
python3 $R/bench.py --steps 20 > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --steps 10 --snr 5 --soft --no-cpu-baseline > $O/bench_soft5db.json 2>> $O/bench.err
python3 $R/bench.py --steps 10 --snr 5 --no-cpu-baseline > $O/bench_hard5db.json 2>> $O/bench.err
python3 $R/bench.py --steps 10 --snr 7 --soft --no-cpu-baseline > $O/bench_soft7db.json 2>> $O/bench.err
python3 $R/bench.py --steps 10 --snr 7 --no-cpu-baseline > $O/bench_hard7db.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --steps 10 > $O/bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find $O/trace -name "*domain_stats.csv" | head -1) $O/bench_domain_stats.csv 2>/dev/null
rm -rf $O/trace
rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -- python3 $R/bench.py --no-cpu-baseline --no-variants --steps 5 > /dev/null 2>> $O/bench.err
python3 $R/tools/timeline.py $O/trace2 > $O/step_timeline.txt      # the default step (no variants after it)
rm -rf $O/trace2
C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
C2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"
rocprofv3 --pmc $C1 --output-format csv -d $O/pmc_f1 -- python3 $R/bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc $C2 --output-format csv -d $O/pmc_f2 -- python3 $R/bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 > /dev/null 2>&1
( echo "# rocprofv3 --pmc <8 SQ counters> --output-format csv -- python3 bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1 (two passes, end of round 2);"
  echo "# counter sums over all dispatches of a kernel in the run (set-up decode + 3 full decodes + the K2 roofline launches); tools/sq_pmc_summary.py"
  python3 $R/tools/sq_pmc_summary.py pmc_f1=$O/pmc_f1 pmc_f2=$O/pmc_f2 ) > $O/sq_pmc_summary.csv
rm -rf $O/pmc_f1 $O/pmc_f2
# K2 HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
for CN in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CN --output-format csv -d $O/pmc_$CN -- python3 $R/bench.py --no-cpu-baseline --no-variants --steps 1 --warmup 1 > /dev/null 2>&1
done
( echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --no-variants --steps 1 --warmup 1; sums per kernel, KiB"
  python3 $R/tools/sq_pmc_summary.py fetch=$O/pmc_FETCH_SIZE write=$O/pmc_WRITE_SIZE | grep -E "pass,|ofdm_fft|ofdm_demap|viterbi_fused" ) > $O/k2_pmc_traffic.csv
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates > $O/valu_rates.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/ubench/hbm_rates.hip -o /tmp/hbm_rates && /tmp/hbm_rates > $O/hbm_rates.txt 2>&1
ls -la $O
