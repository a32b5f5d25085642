for cfg in "192 256 32" "96 128 32" "1 1 32"; do
  set -- $cfg
  MMTTA_SPLIT_BELOW=$1 MMTTA_SPLIT_TARGET=$2 MMTTA_KCI64=$3 timeout -k 10 200 python scripts/layer_times.py > gpurun_out/lt_$1_$2_$3.txt 2>/dev/null
  echo "== below $1 target $2 kci64 $3: $(head -1 gpurun_out/lt_$1_$2_$3.txt)"
done
timeout -k 10 400 python -m pytest tests/test_hip_pointwise.py tests/test_hip_tta.py tests/test_hip_unet.py tests/test_hip_deepfusion.py -m gpu -q -x 2>&1 | tail -3
