# Builds are made on the CPU box (hipcc cross-compiles): see the header of each probe.  One gpurun call:
#   gpurun -- 'bash tools/probes/run_probes.sh'
set -e
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
$R/tools/_chainprobe.bin > $G/r02_mfma_chain_probe.txt
$R/tools/_trail_probe.bin > $G/r02_trail_probe.txt
python3 $R/tools/qrocc.py > $G/r02_qr_occupancy.txt
MPBP_QR_PROF=1 python3 $R/tools/qrphase.py > $G/r02_qr_phase_split.txt 2>&1
tail -3 $G/r02_qr_occupancy.txt
