#!/bin/bash
# tools/pmc_mem.sh TAG [bench.py args...] -- L2 hit/miss and memory-side request counters per kernel (one --pmc pass each), on the GPU box.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_pmc_mem.txt
i=0
for P in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  RH_NO_GRAPH=1 rocprofv3 --pmc $P -d $OUT/${TAG}_mem$i -o ${TAG}_mem$i --output-format csv -- python3 $ROOT/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_mem$i.log 2>&1 || { tail -5 $OUT/${TAG}_mem$i.log; exit 1; }
  python3 $ROOT/tools/pmc_summary.py $OUT/${TAG}_mem$i >> $OUT/${TAG}_pmc_mem.txt
  rm -rf $OUT/${TAG}_mem$i
done
echo "pmc_mem $TAG done"
