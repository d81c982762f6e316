#!/bin/bash
# FETCH_SIZE of k_tower16_bf16 for A/B builds of the library (ab/lib_<name>.so built with SIGMAZERO_EXTRA_FLAGS, see sigma-zero_amd/build.py):
#   bash tools/fetch_ab.sh <tag> new noskip flat flat_noskip        ("new" = the in-tree library)
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@" "$@"; do
    if [ $v = new ]; then unset SIGMAZERO_LIB; else export SIGMAZERO_LIB=$R/ab/lib_$v.so; fi
    rm -rf /tmp/fp_$v
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fp_$v -o run -- python3 $R/tools/tower_pmc.py 4096 bits > /dev/null 2>&1
    echo "== $v" >> $R/gpurun_out/${TAG}_fetch_ab.txt
    python3 $R/tools/pmc_summary.py /tmp/fp_$v k_tower16 >> $R/gpurun_out/${TAG}_fetch_ab.txt
done
cat $R/gpurun_out/${TAG}_fetch_ab.txt
