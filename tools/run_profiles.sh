# Round profiles on one MI355X: bench line, rocprofv3 kernel stats of the same command, and the PMC passes (FETCH_SIZE,
# WRITE_SIZE, matrix-core counters -- each in its own run, never combined with other trace domains).  Usage:
#   bash tools/run_profiles.sh r02b        -> gpurun_out/<tag>_*
set -o pipefail
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${tag}_bench.json 2> >(tee gpurun_out/${tag}_bench.err >&2) || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_stats.err || exit 1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --graph off > /dev/null 2> gpurun_out/${tag}_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --graph off > /dev/null 2> gpurun_out/${tag}_write.err || exit 1
echo "traffic passes done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/${tag}_mfma -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --graph off > /dev/null 2> gpurun_out/${tag}_mfma.err || exit 1
echo "mfma pass done"
