# One gpurun call: bench + rocprofv3 passes + probes of the SHIPPED build of round 4 (profiles/README.md).
# `python tools/profiles.py r04`, `python tools/pmc_cq.py r04 r04 6400x1600x16 16384x4096x1` assemble profiles/r04_*.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $G/r04_cal_$c -o p -- $R/tools/_fetch_probe.bin > $G/r04_cal_$c.log 2>&1
done
python3 $R/bench.py --steps 3 --warmup 3 > $G/r04_bench.json 2> $G/r04_bench.err
python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --phase-profile > $G/r04_bench_phase_timers.json 2> $G/r04_phase.txt
MPBP_SWEEP2=grid python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline > $G/r04_bench_sweep2grid.json 2> $G/r04_bench_sweep2grid.err
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r04_profK -o p -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline > $G/r04_bench_under_rocprof.json 2> $G/r04_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $G/r04_pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline > $G/r04_pmc_$c.json 2> $G/r04_pmc_$c.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $G/r04_pmc_MFMA -o p -- python3 $R/bench.py --steps 1 --warmup 3 --no-cpu-baseline > $G/r04_pmc_MFMA.json 2> $G/r04_pmc_MFMA.err || echo "MFMA pmc pass failed"
# the batched QR: timings, one profiled run (its exit code is the at-exit check of DESIGN 4.7), counter passes of two shapes
cd $R
python3 tools/qrbench3.py > $G/r04_qrbench3.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r04_qr_kt -o p -- python3 tools/qrbench3.py 7200x900x128 > $G/r04_qr_kt.log 2>&1; echo "rocprofv3 --kernel-trace -- python3 tools/qrbench3.py 7200x900x128: exit code $?" > $G/r04_qr_kt.rc
for shape in 6400x1600x16 16384x4096x1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $G/r04_pmc_${shape}_$c -o p -- python3 tools/qrbench3.py $shape > $G/r04_pmc_${shape}_$c.log 2>&1
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $G/r04_pmc_${shape}_MFMA -o p -- python3 tools/qrbench3.py $shape > $G/r04_pmc_${shape}_MFMA.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $G/r04_pmc_${shape}_LDS -o p -- python3 tools/qrbench3.py $shape > $G/r04_pmc_${shape}_LDS.log 2>&1 || echo "LDS pmc pass failed"
done
# probes of the node factorisation, of the fp64 MFMA issue rate and of the update (tools/probes/cq_fac_probe.hip, mfma_operand_probe.hip, cq_upd_probe.hip;
# the first part of profiles/r04_cq_upd_probe.txt - the round-3 tile code - was taken before that code was removed)
( echo "# tools/_cq_fac_new.bin (cq::k_cq_fac2 of the shipped header; -DCQ_PROF build below)"; timeout -k 5 60 tools/_cq_fac_new.bin; timeout -k 5 60 tools/_cq_fac_new_prof.bin
  echo "# tools/_cq_fac_old.bin (the round-3 header: run-time column index)"; timeout -k 5 60 tools/_cq_fac_old.bin ) > $G/r04_cq_fac_probe.txt 2>&1
( echo "# tools/probes/mfma_operand_probe.hip"; timeout -k 5 60 tools/_mfma_operand.bin ) > $G/r04_mfma_operand_probe.txt 2>&1
( echo "# tools/_cq_upd_probe256.bin 256 (cq::k_cq_upd<256> of the shipped header: 256 trailing tiles x 64 nodes)"; timeout -k 5 60 tools/_cq_upd_probe256.bin 256
  echo "# the same, 1024 tiles (four rounds of workgroups at 64 tiles per group)"; timeout -k 5 60 tools/_cq_upd_probe256.bin 1024
  echo "# -DCQ_TRACE, 256 tiles"; timeout -k 5 60 tools/_cq_upd_probe_tr.bin 256 | cut -c1-330
  echo "# -DCQ_NT=512 (two waves per SIMD, tiles through an LDS counter), 256 tiles"; timeout -k 5 60 tools/_cq_upd_probe512.bin 256
  echo "# -DCQ_NT=512, 1024 tiles"; timeout -k 5 60 tools/_cq_upd_probe512.bin 1024 ) > $G/r04_cq_upd_probe_z.txt 2>&1
ls $G/r04_profK $G/r04_pmc_FETCH_SIZE
