#!/bin/bash
# tools/pmc_sq.sh TAG [bench.py args...] -- SQ issue/stall counters per kernel (two --pmc passes, 8 SQ slots each), on the GPU box.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_pmc_sq.txt
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  RH_NO_GRAPH=1 rocprofv3 --pmc $P -d $OUT/${TAG}_sq$i -o ${TAG}_sq$i --output-format csv -- python3 $ROOT/bench.py "$@" --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_sq$i.log 2>&1 || { tail -5 $OUT/${TAG}_sq$i.log; exit 1; }
  python3 $ROOT/tools/pmc_summary.py $OUT/${TAG}_sq$i >> $OUT/${TAG}_pmc_sq.txt
  rm -rf $OUT/${TAG}_sq$i
done
echo "pmc_sq $TAG done"
