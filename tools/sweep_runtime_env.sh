ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT" || exit 1
for spec in "DEBUG_CLR_GRAPH_PACKET_CAPTURE 0 1" "HIP_FORCE_DEV_KERNARG 0 1" "DEBUG_HIP_FORCE_GRAPH_QUEUES 2 4" "AMD_OPT_FLUSH 0 1" "GPU_STREAMOPS_CP_WAIT 0 1" "DEBUG_HIP_GRAPH_BATCH_SIZE 1 64"; do
  set -- $spec
  bash tools/ab_bench.sh $1 $2 $3 2
done
