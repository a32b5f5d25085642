#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2s; mkdir -p $O; cd $R
python scripts/layer_times.py > $O/layers.txt 2>/dev/null; head -1 $O/layers.txt
grep "igemm" $O/layers.txt | sort -k1 -n -r | head -12
b() { env $1 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])"; }
for i in 1 2; do
  echo "default (class-fused >= 128, vec16 epilogue): $(b MMTTA_X=0)"
  echo "class-fused off:                              $(b MMTTA_CLSFUSE=0)"
  echo "vec16 epilogue off:                           $(b MMTTA_NO_EPIVEC=1)"
  echo "class-fused >= 32:                            $(b MMTTA_CLSFUSE=32)"
done
