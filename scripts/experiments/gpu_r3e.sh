set -o pipefail
O=gpurun_out/r3e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/all.log 2>&1; tail -5 $O/all.log; grep "fullsize" $O/all.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3e/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','lanes_equal','post_tta_dice') if k in d}); print(d.get('variants')); print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us']); print(d.get('parity_full_size'))
PY
