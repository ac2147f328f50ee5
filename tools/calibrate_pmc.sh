#!/bin/bash
# tools/calibrate_pmc.sh TAG -- on the GPU box: run tools/probes/fetch_calib.bin (known byte counts, buffer 8 x the Infinity Cache) under
# two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) and print bytes-per-counter-unit for every access pattern
# -> gpurun_out/TAG_pmc_calibration.txt / .json (copy into profiles/; tools/pmc_traffic_json.py reads profiles/pmc_calibration.json).
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
BIN=$ROOT/tools/probes/fetch_calib.bin
[ -x $BIN ] || hipcc --offload-arch=gfx950 -O3 -o $BIN $ROOT/tools/probes/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
$BIN > $OUT/${TAG}_calib_bytes.txt || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/${TAG}_calib_$C -o ${TAG}_calib_$C --output-format csv -- $BIN > $OUT/${TAG}_calib_$C.log 2>&1 || { tail -5 $OUT/${TAG}_calib_$C.log; exit 1; }
done
python3 $ROOT/tools/pmc_calibrate.py $OUT/${TAG}_calib_bytes.txt $OUT/${TAG}_calib_FETCH_SIZE $OUT/${TAG}_calib_WRITE_SIZE $OUT/${TAG}_pmc_calibration.json | tee $OUT/${TAG}_pmc_calibration.txt
rm -rf $OUT/${TAG}_calib_FETCH_SIZE $OUT/${TAG}_calib_WRITE_SIZE
