#!/bin/bash
# Extra GPU checks of a round: smoke(), the other bench configurations (each with its roofline block), whole-volume PMC.
set -e -o pipefail
out=gpurun_out/${1:-extras}
mkdir -p $out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
run() {   # name, bench arguments...
  local name=$1; shift
  timeout -k 10 500 python bench.py "$@" --no-variants > $out/bench_$name.json 2> $out/bench_$name.err || { tail -20 $out/bench_$name.err; exit 1; }
  python -c "import json;d=json.load(open('$out/bench_$name.json'));r=d['roofline'];print('$name', round(d['value'],2), 'vol/s', round(d['ms_per_step'],2), 'ms;', r['kernel'], 'frac', round(r['frac'],3), '| whole volume', {k: round(v,3) for k,v in r['whole_volume'].items() if k.startswith('frac') or k.startswith('eff')})"
}
# (whole rounds of lanes x group volumes: a partial round runs smaller groups)
run hecktor_unet --task hecktor21 --steps 48 --warmup 8
run deepfusion_brats --model unet_multimodal_deepfusion --steps 24 --warmup 4
run unet_fp32 --precision fp32 --steps 24 --warmup 4
run brats_full_160x192x160 --shape 160 192 160 --steps 24 --warmup 4
# BASELINE configs[4] on one GPU: missing modality + modality dropout (the U-Net with the channel zeroed, the deep-fusion net
# with its branch masked out of the means)
run moddrop_unet --method tta_moddrop --steps 48 --warmup 8
run moddrop_deepfusion --method tta_moddrop --model unet_multimodal_deepfusion --steps 24 --warmup 4
if [ "$2" != "nopmc" ]; then
  bash scripts/pmc_bench.sh > $out/pmc.log 2>&1 || { tail -20 $out/pmc.log; exit 1; }
  cp gpurun_out/pmc_bench/sq_per_kernel.txt gpurun_out/pmc_bench/mem_per_kernel.txt gpurun_out/pmc_bench/traffic.json $out/
fi
