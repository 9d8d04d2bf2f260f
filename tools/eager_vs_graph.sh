# The bench step as a hipGraph replay, eager, and eager with the data-parallel reducer forced on one rank (what N > 1 runs):
#   bash tools/eager_vs_graph.sh
run() { "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LABEL', d['value'], d['ms_per_step'])"; }
LABEL="graph      " run python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --graph auto
LABEL="eager      " run python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --graph off
LABEL="eager + dp " run env CNK_DP_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --graph off
