# configs[3] whole-graph sweeps with the batched sweeps' section timing and a clock / temperature / power sample every 15 s
# (is the slower fourth sweep a property of the work or of the chip's clock?)
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out
( while true; do echo "t=$(date +%s) $(rocm-smi --showclocks --showtemp --showpower 2>/dev/null | grep -E 'sclk|Temperature \(Sensor junction\)|Average Graphics Package Power|Current Socket' | tr -s ' ' | tr '\n' ';')"; sleep 15; done ) > $G/r03_config3_clocks.txt 2>&1 &
MON=$!
MPBP_V2_TIMING=1 timeout -k 5 1000 python $R/tools/bigconfigs.py karate 200 40 4 > $G/r03_config3_timing.log 2>&1
kill $MON
grep -c "v2 timing" $G/r03_config3_timing.log
grep "karate" $G/r03_config3_timing.log
