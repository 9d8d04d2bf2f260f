# rocprofv3 kernel stats of the bench step under two settings of one environment switch (A/B):
#   bash tools/ab_stats.sh CNK_BN_FOLD 0 1 [tag]     -> gpurun_out/<tag>_<value>_stats/..., summarised by tools/ab_stats.py
set -o pipefail
[ $# -ge 3 ] || { echo "usage: ab_stats.sh VAR A B [tag]" >&2; exit 2; }
var=$1; a=$2; b=$3; tag=${4:-ab}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp; cd "$ROOT" || { echo "ab_stats.sh: cannot cd to $ROOT" >&2; exit 1; }
mkdir -p gpurun_out
for v in $a $b; do
  export $var=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${v}_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_${v}.json 2> gpurun_out/${tag}_${v}.err || { tail -20 gpurun_out/${tag}_${v}.err >&2; exit 1; }
  echo "$var=$v done"
done
