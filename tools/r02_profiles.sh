set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
python3 $R/bench.py --steps 2 --warmup 3 > $G/r02_bench.json 2> $G/r02_bench.err
python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --phase-profile > $G/r02_bench_phase_timers.json 2> $G/r02_phase.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $G/profK -o p -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline > $G/r02_bench_under_rocprof.json 2> $G/r02_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $G/pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline > $G/pmc_$c.json 2> $G/pmc_$c.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $G/pmc_MFMA -o p -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline > $G/pmc_MFMA.json 2> $G/pmc_MFMA.err || echo "MFMA pmc pass failed"
ls $G/profK $G/pmc_FETCH_SIZE $G/pmc_MFMA
