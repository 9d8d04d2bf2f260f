# A/B of the bench step under two settings of one environment switch, alternating, N pairs (box-to-box spread is as large as
# most single changes, so both settings run in ONE call):   bash tools/ab_bench.sh CNK_BN_FOLD 0 1 3 [bench.py args]
[ $# -ge 3 ] || { echo "usage: ab_bench.sh VAR A B [pairs] [bench.py args]" >&2; exit 2; }
var=$1; a=$2; b=$3; n=${4:-3}; shift 3; [ $# -gt 0 ] && shift
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT" || exit 1
mkdir -p gpurun_out
for i in $(seq 1 $n); do
  for v in $a $b; do
    env $var=$v python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], d['ms_per_step'], d['roofline'].get('wgrad_share',{}).get('chosen'))"
  done
done
