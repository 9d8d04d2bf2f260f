# Per-kernel averages of the once-per-step (stem / pool / optimizer) kernels of the bench step, from a rocprofv3 kernel trace.
#   bash tools/stem_stats.sh <tag>      -> gpurun_out/<tag>_stem.txt
set -o pipefail
tag=${1:-stem}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_stats.err || exit 1
f=$(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | sort | tail -1)
python3 - "$f" > gpurun_out/${tag}_stem.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 7
for r in rows:
    per_step = float(r['TotalDurationNs']) / steps / 1e3
    if per_step > 60:
        print('%-90s %6s calls  avg %8.1f us  %7.1f us/step' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, per_step))
print('total %.2f ms/step' % (tot / steps / 1e6))
P
cat gpurun_out/${tag}_stem.txt
