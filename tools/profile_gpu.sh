#!/bin/bash
# tools/profile_gpu.sh TAG KEY [bench.py args...] -- run on the GPU box (through gpurun) from the repository root.
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py ARGS --isolate` -> gpurun_out/TAG_kernel_stats.txt (+ the bench line).
#    --isolate: the sweeps of a step run one after the other, so the trace's per-kernel averages are the isolated durations that
#    bench.py's live event timing reports (roofline.avg_launch_us); without it the duplex stream overlaps the McCaskill stream.
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md "PMC slots") over the fixed
#    workload tools/pmc_workload.py with RH_NO_GRAPH=1 (counters attach to host-launched dispatches) -> gpurun_out/TAG_pmc_traffic.txt,
#    folded into profiles/pmc_traffic.json under KEY (= <model>_n<N>_b<pairs per step>) by tools/pmc_traffic_json.py.
# Copy what should be judged into profiles/ afterwards (profiles/pmc_traffic.json is written in place: copy it back from gpurun_out/).
set -o pipefail
TAG=$1; KEY=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
MODEL=$(echo $KEY | sed 's/_n.*//'); N=$(echo $KEY | sed 's/.*_n\([0-9]*\)_b.*/\1/'); BATCH=$(echo $KEY | sed 's/.*_b\([0-9]*\).*/\1/')
HP=""; case "$KEY" in *duplex*) HP="--hp duplex";; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o ${TAG} -- python3 $ROOT/bench.py "$@" --isolate --no-cpu-baseline > $OUT/${TAG}_bench.log 2>&1 || { tail -5 $OUT/${TAG}_bench.log; exit 1; }
grep '^{' $OUT/${TAG}_bench.log | tail -1 > $OUT/${TAG}_bench_under_rocprof.json
DB=$(ls $OUT/${TAG}_trace/*/*.db $OUT/${TAG}_trace/*.db 2>/dev/null | head -1)
python3 $ROOT/tools/rocprof_summary.py "$DB" "bench.py $* --isolate --no-cpu-baseline" > $OUT/${TAG}_kernel_stats.txt
rm -rf $OUT/${TAG}_trace
: > $OUT/${TAG}_pmc_traffic.txt
COMPUTES=3
for C in FETCH_SIZE WRITE_SIZE; do
  RH_NO_GRAPH=1 rocprofv3 --pmc $C -d $OUT/${TAG}_pmc_$C -o ${TAG}_$C --output-format csv -- python3 $ROOT/tools/pmc_workload.py --model $MODEL --n $N --batch $BATCH $HP --computes $((COMPUTES-1)) > $OUT/${TAG}_pmc_$C.log 2>&1 || { tail -5 $OUT/${TAG}_pmc_$C.log; exit 1; }
  python3 $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_$C >> $OUT/${TAG}_pmc_traffic.txt
done
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE   # raw CSVs are large; the summary stays
python3 $ROOT/tools/pmc_traffic_json.py $KEY $OUT/${TAG}_pmc_traffic.txt $COMPUTES && cp $ROOT/profiles/pmc_traffic.json $OUT/pmc_traffic.json
echo "profile $TAG done"
