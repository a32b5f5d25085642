#!/bin/bash
# Same-box ablation of the final build: bench.py (unet 4x128^3, S = 10, bf16) with one switch changed at a time.
R=$GRAFT_REPO_ROOT; cd $R
b() { env $1 python bench.py $2 --steps 24 --warmup 4 --no-cpu-baseline --no-profile-pass --no-variants 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['value'])"; }
for rep in 1 2; do
  echo "rep $rep  default build                                      $(b MMTTA_X=1)"
  echo "rep $rep  row loader + stage prefetch off (option 6 = 0)     $(b MMTTA_NO_PIPE=1)"
  echo "rep $rep  16-byte epilogue off (option 9 = 0)                $(b MMTTA_NO_EPIVEC=1)"
  echo "rep $rep  class-fused stride-2 forms off (option 12 = 0)     $(b MMTTA_CLSFUSE=0)"
  echo "rep $rep  round-1 weight-gradient kernels (option 11 = 0)    $(b MMTTA_WGVEC=0)"
  echo "rep $rep  fp32 storage of the activations                    $(b MMTTA_X=1 '--storage fp32')"
  echo "rep $rep  3->3 layers on the vector ALU (option 13 = 0)      $(b MMTTA_THINMFMA=0)"
  echo "rep $rep  3->3 matrix tiles, 8x8x64 tile (option 13 = 1)     $(b MMTTA_THINMFMA=1)"
  echo "rep $rep  8x8x8 tile for the 32-column layers (option 10 = 0) $(b MMTTA_LEAN=0)"
  echo "rep $rep  three-pass norm backward at the 8^3 levels         $(b MMTTA_NORM_SMALL=0)"
  for l in 1 2 3; do echo "rep $rep  lanes $l                                            $(b MMTTA_X=1 "--lanes $l")"; done
done
