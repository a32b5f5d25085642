#!/bin/bash
# Extra GPU checks of a round: smoke(), the HECKTOR-shaped and deep-fusion bench configurations, PMC passes.
set -e -o pipefail
out=gpurun_out/${1:-extras}
mkdir -p $out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
timeout -k 10 400 python bench.py --task hecktor21 --steps 4 --warmup 1 --no-cpu-baseline --no-profile-pass > $out/bench_hecktor.json 2> $out/bench_hecktor.err || { tail -20 $out/bench_hecktor.err; exit 1; }
python -c "import json;d=json.load(open('$out/bench_hecktor.json'));print('hecktor unet', d['value'], d['ms_per_step'], d['config']['workload'][:60])"
timeout -k 10 500 python bench.py --model unet_multimodal_deepfusion --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass > $out/bench_deepfusion.json 2> $out/bench_deepfusion.err || { tail -20 $out/bench_deepfusion.err; exit 1; }
python -c "import json;d=json.load(open('$out/bench_deepfusion.json'));print('deepfusion brats', d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --precision fp32 --steps 4 --warmup 1 --no-cpu-baseline --no-profile-pass > $out/bench_fp32.json 2> $out/bench_fp32.err || { tail -20 $out/bench_fp32.err; exit 1; }
python -c "import json;d=json.load(open('$out/bench_fp32.json'));print('unet fp32', d['value'], d['ms_per_step'])"
bash scripts/pmc_layers.sh > $out/pmc.log 2>&1 || { tail -20 $out/pmc.log; exit 1; }
python scripts/pmc_summary.py gpurun_out/pmc_lt/p1/run_results.db gpurun_out/pmc_lt/p2/run_results.db gpurun_out/pmc_lt/p3/run_results.db gpurun_out/pmc_lt/p4/run_results.db --json $out/pmc_traffic.json > $out/pmc_summary.txt
head -5 $out/pmc_summary.txt | cut -c1-200
