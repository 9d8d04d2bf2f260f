# Kernel timeline of the captured bench step (tools/trace_gaps.py over a rocprofv3 --kernel-trace of bench.py):
#   bash tools/timeline.sh tag [ENV=VALUE ...]      -> gpurun_out/<tag>_timeline.txt   (SKIP=0: the last, eager, step)
tag=${1:-tl}; [ $# -gt 0 ] && shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp; cd "$ROOT" || { echo "timeline.sh: cannot cd to $ROOT" >&2; exit 1; }
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_trace_bench.json 2> gpurun_out/${tag}_trace.err || { tail -20 gpurun_out/${tag}_trace.err >&2; exit 1; }
f=$(find gpurun_out/${tag}_trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_gaps.py "$f" ${SKIP:-6} > gpurun_out/${tag}_timeline.txt
rm -rf gpurun_out/${tag}_trace
cat gpurun_out/${tag}_timeline.txt
