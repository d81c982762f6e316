#!/bin/bash
# effective shader clock under load (MI355X_MICROARCH.md "DVFS give-back"): GRBM_GUI_ACTIVE / 8 / kernel wall time
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/clk && mkdir -p /tmp/clk
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk/tower -o run -- python3 $ROOT/tools/tower_pmc.py 4096 > /dev/null 2> /tmp/clk/err1.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk/block -o run -- python3 $ROOT/tools/block_pmc.py > /dev/null 2> /tmp/clk/err2.txt
python3 - <<'PY'
import csv, glob, collections
for tag, pat in (("tower", "k_tower16"), ("block", "k_block16")):
    cnt = {}
    for f in glob.glob("/tmp/clk/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[r["Dispatch_Id"]] = (float(r["Counter_Value"]), r)
    dur = {}
    for f in glob.glob("/tmp/clk/%s/**/*kernel_trace.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    rows = [(cnt[k][0], dur[k]) for k in cnt if k in dur]
    rows = rows[len(rows) // 3:]
    for c, d in rows[:4]:
        print(tag, "GRBM_GUI_ACTIVE %.4g  wall %.1f us  clock %.3f GHz" % (c, d * 1e6, c / 8 / d / 1e9))
PY
