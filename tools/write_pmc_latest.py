"""After `bash tools/profile_round.sh <tag>` (on the GPU box) has merged its summaries into gpurun_out/: copy them to profiles/ and rewrite
profiles/pmc_tower_latest.json / pmc_tree_latest.json (what bench.py quotes as `roofline.traffic`) with the hash of the kernel sources the counters were
collected from.  bench.py reports traffic = null when the sources have changed since."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
src_hash = json.load(open(os.path.join(G, tag + "_source_hash.json")))


def counters(path, kernel):
    out = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+(\S+)\s+n=(\d+) mean=(\S+)", line)
        if m and m.group(1) == kernel:
            out[m.group(2)] = float(m.group(4))
    return out


for f in os.listdir(G):
    if f.startswith(tag + "_") and (f.endswith(".txt") or f.endswith(".csv") or f.endswith(".json")):
        shutil.copy(os.path.join(G, f), os.path.join(P, f))
t = counters(os.path.join(G, tag + "_pmc_k_tower16_B4096.txt"), "k_tower16")
latest = json.load(open(os.path.join(P, "pmc_tower_latest.json")))
latest.update({"source": "profiles/%s_pmc_k_tower16_B4096.txt (rocprofv3 --pmc, separate passes; tools/profile_round.sh %s)" % (tag, tag),
               "source_hash": src_hash["tower"], "FETCH_SIZE_KB_raw": t["FETCH_SIZE"], "WRITE_SIZE_KB": t["WRITE_SIZE"],
               "hbm_bytes_per_launch": int(2 * t["FETCH_SIZE"] * 1024 + t["WRITE_SIZE"] * 1024)})
json.dump(latest, open(os.path.join(P, "pmc_tower_latest.json"), "w"), indent=1)
tr = counters(os.path.join(G, tag + "_pmc_tree_kernels_B4096.txt"), "k_search_step")
lt = json.load(open(os.path.join(P, "pmc_tree_latest.json")))
lt.update({"source_hash": src_hash["tree"], "hbm_bytes_per_launch": int(2 * tr["FETCH_SIZE"] * 1024 + tr["WRITE_SIZE"] * 1024),
           "source": "profiles/%s_pmc_tree_kernels_B4096.txt (tools/profile_round.sh %s)" % (tag, tag)})
json.dump(lt, open(os.path.join(P, "pmc_tree_latest.json"), "w"), indent=1)
sp = os.path.join(G, tag + "_pmc_k_tower_split_B4096.txt")
if os.path.exists(sp):
    c = counters(sp, "k_tower_split")
    json.dump({"kernel": "k_tower_split<2>, B=4096, bit-packed planes (tools/tower_pmc.py 4096 bits split)", "source_hash": src_hash["split"],
               "source": "profiles/%s_pmc_k_tower_split_B4096.txt" % tag, "counters": c,
               "hbm_bytes_per_launch": int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024),
               "correction": "FETCH_SIZE x2 (calibrated, profiles/r02r_fetch_calib.txt); counts L2 misses served by the Infinity Cache as well as HBM"},
              open(os.path.join(P, "pmc_split_latest.json"), "w"), indent=1)
print("profiles/pmc_*_latest.json rewritten for source hash", src_hash)
