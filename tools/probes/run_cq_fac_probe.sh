# builds tools/_cq_fac_probe_<n>.bin for the variants of the column step (run on the GPU box: for v in 0 1 2 3 4 5; do tools/_cq_fac_probe_$v.bin; done)
set -e
cd "$(dirname "$0")/../.."
for v in 0 1 2 3 4 5; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-result -Wno-unused-value -DCQ_VAR=$v -Imatrixproductbp.jl_amd/csrc tools/probes/cq_fac_probe.hip -o tools/_cq_fac_probe_$v.bin
done
