#!/bin/bash
# round 3: optimizer with non-temporal loads / stores of the gradient and the moments (same-box A/B under the kernel trace)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6n; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for nt in 0 1 0 1; do
  export MMTTA_OPTIM_NT=$nt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$nt -o run -- python3 $R/bench.py --steps 24 --warmup 24 --no-cpu-baseline --no-variants --no-profile-pass > $O/b.json 2> $O/err.txt
  python3 -c "
import csv,glob,json
d=json.load(open('$O/b.json')); f=glob.glob('$O/p$nt/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'optim_kernel' in r['Name'] or 'pack_batched' in r['Name']: print('nt=$nt', r['Name'][:40], r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
print('nt=$nt bench', round(d['value'],2))"
  rm -rf $O/p$nt
done
