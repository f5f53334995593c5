# One gpurun call: bench + rocprofv3 passes of the SHIPPED build (profiles/README.md).  tools/profiles.py r03 assembles profiles/r03_*.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
# counter calibration for the engine's access shapes (8 / 16 / 32 bytes per lane), program directly after `--`
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $G/r03_cal_$c -o p -- $R/tools/_fetch_probe.bin > $G/r03_cal_$c.log 2>&1
done
python3 $R/bench.py --steps 2 --warmup 3 > $G/r03_bench.json 2> $G/r03_bench.err
python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --phase-profile > $G/r03_bench_phase_timers.json 2> $G/r03_phase.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r03_profK -o p -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline > $G/r03_bench_under_rocprof.json 2> $G/r03_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $G/r03_pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline > $G/r03_pmc_$c.json 2> $G/r03_pmc_$c.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $G/r03_pmc_MFMA -o p -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline > $G/r03_pmc_MFMA.json 2> $G/r03_pmc_MFMA.err || echo "MFMA pmc pass failed"
ls $G/r03_profK $G/r03_pmc_FETCH_SIZE $G/r03_cal_FETCH_SIZE
