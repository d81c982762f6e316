#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench workload (1 warm-up + 1 timed ply)  -> gpurun_out/<tag>_kernel_stats.csv
#   2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ | GRBM) of the tree kernels, the bf16 tower and the split-precision tower -> gpurun_out/<tag>_pmc_*.txt
#      (SKIP_BLOCK=0 adds the per-block kernel); afterwards, locally:  python tools/write_pmc_latest.py <tag>  (profiles/pmc_*_latest.json with the source hash)
# Traces are written under /tmp (they exceed what gpurun copies back); only the summaries are kept.
set -e -o pipefail
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import json, sigma_zero_amd.build as b; print(json.dumps({g: b.source_hash(g) for g in ('tower', 'split', 'tree')}))" > $OUT/${TAG}_source_hash.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/stats -o run -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-config --no-train-step --split-plies 1 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err
cp $(find /tmp/prof_$TAG/stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
echo "stats done"
for spec in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"; do
    name=$(echo $spec | cut -d' ' -f1)
    rocprofv3 --pmc $spec --output-format csv -d /tmp/prof_$TAG/tree_$name -o run -- python3 $ROOT/tools/tree_pmc.py bits128 > /dev/null 2>> $OUT/${TAG}_pmc.err
    echo "== $spec" >> $OUT/${TAG}_pmc_tree_kernels_B4096.txt
    python3 $ROOT/tools/pmc_summary.py /tmp/prof_$TAG/tree_$name k_search_step k_search_begin k_play >> $OUT/${TAG}_pmc_tree_kernels_B4096.txt
    echo "tree $name done"
done
for spec in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
    name=$(echo $spec | cut -d' ' -f1)
    rocprofv3 --pmc $spec --output-format csv -d /tmp/prof_$TAG/tower_$name -o run -- python3 $ROOT/tools/tower_pmc.py 4096 bits > /dev/null 2>> $OUT/${TAG}_pmc.err
    echo "== $spec" >> $OUT/${TAG}_pmc_k_tower16_B4096.txt
    python3 $ROOT/tools/pmc_summary.py /tmp/prof_$TAG/tower_$name k_tower16 >> $OUT/${TAG}_pmc_k_tower16_B4096.txt
    echo "tower $name done"
done
for spec in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
    name=$(echo $spec | cut -d' ' -f1)
    rocprofv3 --pmc $spec --output-format csv -d /tmp/prof_$TAG/split_$name -o run -- python3 $ROOT/tools/tower_pmc.py 4096 bits split > /dev/null 2>> $OUT/${TAG}_pmc.err
    echo "== $spec" >> $OUT/${TAG}_pmc_k_tower_split_B4096.txt
    python3 $ROOT/tools/pmc_summary.py /tmp/prof_$TAG/split_$name k_tower_split >> $OUT/${TAG}_pmc_k_tower_split_B4096.txt
    echo "split $name done"
done
[ "${SKIP_BLOCK:-1}" = "1" ] && exit 0
for spec in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    name=$(echo $spec | cut -d' ' -f1)
    rocprofv3 --pmc $spec --output-format csv -d /tmp/prof_$TAG/block_$name -o run -- python3 $ROOT/tools/block_pmc.py > /dev/null 2>> $OUT/${TAG}_pmc.err
    echo "== $spec" >> $OUT/${TAG}_pmc_k_block16_B4096.txt
    python3 $ROOT/tools/pmc_summary.py /tmp/prof_$TAG/block_$name k_block16 >> $OUT/${TAG}_pmc_k_block16_B4096.txt
    echo "block $name done"
done
