set -e
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out; B=$R/tools/_caqr_probe.bin
{
timeout -k 5 60 $B 256 64 1 8 || echo "rc=$?"
timeout -k 5 60 $B 1600 400 1 2 || echo "rc=$?"
timeout -k 5 60 $B 1600 400 256 2 || echo "rc=$?"
} > $G/caqr2.log 2>&1
tail -30 $G/caqr2.log
