#!/bin/bash
# rocprofv3 kernel stats of one training-step run per model family: top kernels and any torch-native (at::native)
# kernels that still run inside the step.   tools/family_profile.sh "<family kind size classes batch dtype>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' .' '__')
  STEPS=5 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fam_$tag -- python3 $R/tools/bench_model.py $cfg > $R/gpurun_out/fam_$tag.log 2>&1
  grep -h "ms/step" $R/gpurun_out/fam_$tag.log
  python3 - "$R/gpurun_out/fam_$tag" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
nat = [r for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name']]
print('   native/copy kernels: %.1f%% of kernel time' % (100 * sum(float(r['TotalDurationNs']) for r in nat) / tot))
for r in rows[:7]:
    print('   %-90s %5s calls %6.1f%%' % (r['Name'][:90], r['Calls'], float(r['Percentage'])))
for r in nat[:4]:
    print('   [native] %-80s %5s calls %6.1f%%' % (r['Name'][:80], r['Calls'], float(r['Percentage'])))
PY
done
