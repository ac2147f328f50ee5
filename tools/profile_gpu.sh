#!/bin/bash
# tools/profile_gpu.sh TAG [bench.py args...] -- run on the GPU box (through gpurun) from the repository root.
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py ARGS` -> gpurun_out/TAG_kernel_stats.txt (+ the bench line)
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md "PMC slots")
#    with RH_NO_GRAPH=1 (counters attach to host-launched dispatches) -> gpurun_out/TAG_pmc_traffic.txt
# Copy what should be judged into profiles/ afterwards.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o ${TAG} -- python3 $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/${TAG}_bench.log 2>&1 || { tail -5 $OUT/${TAG}_bench.log; exit 1; }
grep '^{' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
DB=$(ls $OUT/${TAG}_trace/*/*.db $OUT/${TAG}_trace/*.db 2>/dev/null | head -1)
python3 $ROOT/tools/rocprof_summary.py "$DB" "bench.py $* --no-cpu-baseline" > $OUT/${TAG}_kernel_stats.txt
: > $OUT/${TAG}_pmc_traffic.txt
for C in FETCH_SIZE WRITE_SIZE; do
  RH_NO_GRAPH=1 rocprofv3 --pmc $C -d $OUT/${TAG}_pmc_$C -o ${TAG}_$C --output-format csv -- python3 $ROOT/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_$C.log 2>&1 || { tail -5 $OUT/${TAG}_pmc_$C.log; exit 1; }
  python3 $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_$C >> $OUT/${TAG}_pmc_traffic.txt
done
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE   # raw CSVs are large; the summary stays
echo "profile $TAG done"
