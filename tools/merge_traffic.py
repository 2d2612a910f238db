#!/usr/bin/env python3
"""profiles/traffic.json for bench.py from the passes of one tag:
     gpurun_out/prof_<tag>/traffic.json        tools/pmc_summary.py <tag>          the codec kernels (262 144 chains per launch)
     gpurun_out/prof_<tag>p/traffic.json       tools/pmc_summary.py <tag>p <R>     the text kernels (k_ingest_*, k_pdb_*: 65 536 chains)
     <probe.json>                               tools/hbm_busy_probe.py             memory-controller-side activity per stage
   usage: tools/merge_traffic.py <tag> <probe.json> [<text tag, default <tag>p>]   -> profiles/traffic.json"""
import json, os, sys
tag, probe_path = sys.argv[1], sys.argv[2]
ttag = sys.argv[3] if len(sys.argv) > 3 else tag + "p"
t = json.load(open(os.path.join("gpurun_out", "prof_" + tag, "traffic.json")))
t["source"] = t["source"].replace("profiles/%s_pmc_per_kernel.csv" % tag, "profiles/%s_pmc_per_kernel.csv" % tag)
pdir = os.path.join("gpurun_out", "prof_" + ttag, "traffic.json")
if os.path.exists(pdir):
    tp = json.load(open(pdir))
    for k, v in tp["kernels"].items():
        if k.startswith(("k_ingest", "k_pdb")):
            v["residues_per_launch"] = 22937600
            t["kernels"][k] = v
    t["text_kernels_source"] = "profiles/%s_pmc_per_kernel.csv (the same passes with --pdb-sample 65536: 22 937 600 residues of PDB / mmCIF text per launch)" % ttag
pr = json.load(open(probe_path))
if "probes" in pr:
    ks = t["kernels"]
    tot = lambda k: ks[k]["fetch_bytes_per_residue"] + ks[k]["write_bytes_per_residue"]
    sizes = sum(tot(k) for k in ("k_entry_sizes", "k_sizes_reduce", "k_sizes_mid", "k_sizes_apply") if k in ks)   # every probed decompress call runs the sizes pass too
    cs = {"source": "profiles/%s (tools/hbm_busy_probe.py: amdgpu mem_busy_percent, the memory controllers' activity level, "
                    "calibrated on device copies at 100 / 50 / 25 %% duty: %s of linear)" % (os.path.basename(probe_path), pr.get("linearity")),
          "bytes_per_s_per_percent": pr["bytes_per_s_per_percent"], "sizes_pass_bytes_per_residue_subtracted": round(sizes, 2), "bytes_per_residue": {}}
    for k in ("k_backbone", "k_res_index", "k_sidechain"):
        p = pr["probes"].get(k)
        if p and "hbm_bytes_per_residue_estimate" in p:
            cs["bytes_per_residue"][k] = round(p["hbm_bytes_per_residue_estimate"] - sizes, 1)
    for k in ("decompress_all", "compress_all"):
        p = pr["probes"].get(k)
        if p and "hbm_bytes_per_residue_estimate" in p:
            cs["bytes_per_residue"][k] = p["hbm_bytes_per_residue_estimate"]
    cs["fabric_side_counters_same_scope"] = {"k_backbone": round(tot("k_backbone"), 1), "k_res_index": round(tot("k_res_index"), 1), "k_sidechain": round(tot("k_sidechain"), 1),
                                             "decompress_all": round(sizes + tot("k_backbone") + tot("k_res_index") + tot("k_sidechain"), 1),
                                             "compress_all": round(sum(tot(k) for k in ("k_compress_sizes", "k_compress_index", "k_compress_angles_w", "k_compress_angles", "k_compress_pack") if k in ks), 1)}
    t["controller_side"] = cs
# the mixed-length workload (bench.py --mixed: 542 k chains of the log-normal generator), per residue of THAT batch
mtag = os.environ.get("MIXED_TAG")
if mtag and os.path.exists(os.path.join("gpurun_out", "prof_" + mtag, "traffic.json")):
    tm = json.load(open(os.path.join("gpurun_out", "prof_" + mtag, "traffic.json")))
    assert tm.get("csrc_sha16") == t.get("csrc_sha16"), "the mixed passes describe other kernels than the uniform ones"
    t["mixed"] = {"source": tm["source"], "kernels": {k: v for k, v in tm["kernels"].items() if k.startswith(("k_compress", "k_backbone", "k_res_index", "k_sidechain", "k_entry", "k_sizes"))}}
json.dump(t, open(os.path.join("profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(t.get("controller_side"), indent=1))
