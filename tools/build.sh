#!/bin/bash
# rebuild the HIP library in-tree (what __graft_entry__.build() does), from any cwd
cd "$(dirname "$0")/.." && python -c "import mpbp_amd; mpbp_amd.build()" 2>&1 | tail -3
