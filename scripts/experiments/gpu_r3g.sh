#!/bin/bash
# round 3: PMC traffic of the grouped arrangement (one lane x group 8 under the headline geometry)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
bash scripts/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1 || tail -20 gpurun_out/pmc_bench.log
python scripts/traffic_rank.py gpurun_out/pmc_bench/mem_per_kernel.txt 24 > gpurun_out/pmc_bench/traffic_per_volume.md
head -40 gpurun_out/pmc_bench/traffic_per_volume.md
