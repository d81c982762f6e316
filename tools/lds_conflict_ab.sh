#!/bin/bash
# Where do the tower's LDS bank conflicts come from, and what do they cost?  (run through gpurun from the repo root; needs ab/libsz_noconf.so =
# a build with SIGMAZERO_EXTRA_FLAGS=-DNN_EPI_NOCONFLICT=1, timing only)
#   1. counters of the shipped kernel, of its stamped build, and of the stamped build WITHOUT the K loop's ds_read_b128 (mode 3): what is left is the epilogue
#   2. counters + in-kernel block cycles with the epilogue's stores and residual reads moved to conflict-free addresses
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r03g}_lds_conflict_ab.txt
cd /tmp && export TMPDIR=/tmp
pmc() {  # $1 = label, $2.. = tower_pmc.py args ; env SIGMAZERO_LIB honoured
    rm -rf /tmp/ldsab && mkdir -p /tmp/ldsab
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ldsab/t -o run -- python3 $ROOT/tools/tower_pmc.py "${@:2}" > /dev/null 2> /tmp/ldsab/err.txt
    echo "== $1" >> $OUT
    python3 $ROOT/tools/pmc_summary.py /tmp/ldsab/t k_tower16 >> $OUT
}
: > $OUT
pmc "shipped kernel" 4096 bits bf16
pmc "stamped build, full K loop (mode 1)" 4096 bits bf16 1
pmc "stamped build, no LDS fragment reads in the K loop (mode 3): epilogue + staging only" 4096 bits bf16 3
export SIGMAZERO_LIB=$ROOT/ab/libsz_noconf.so
pmc "NN_EPI_NOCONFLICT build (epilogue slots conflict-free, timing only)" 4096 bits bf16
unset SIGMAZERO_LIB
echo "== in-kernel block cycles, shipped" >> $OUT
python3 $ROOT/tools/tower_stamps.py 4096 1 >> $OUT 2>&1
echo "== in-kernel block cycles, NN_EPI_NOCONFLICT build" >> $OUT
SIGMAZERO_LIB=$ROOT/ab/libsz_noconf.so python3 $ROOT/tools/tower_stamps.py 4096 1 >> $OUT 2>&1
echo "== again shipped (same box, drift check)" >> $OUT
python3 $ROOT/tools/tower_stamps.py 4096 1 >> $OUT 2>&1
cat $OUT
