#!/bin/bash
# LDS bank-conflict share of the persistent tower: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (rocprofv3 --pmc, own pass)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ldspmc && mkdir -p /tmp/ldspmc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ldspmc/t -o run -- python3 $ROOT/tools/tower_pmc.py 4096 bits > /dev/null 2> /tmp/ldspmc/err.txt
python3 $ROOT/tools/pmc_summary.py /tmp/ldspmc/t k_tower16
