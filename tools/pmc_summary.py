#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/pmc_*_per_kernel.csv: per-kernel issue/stall shares and traffic per residue."""
import collections, csv, glob, os, sys
tag = sys.argv[1]; R = float(sys.argv[2]) if len(sys.argv) > 2 else 262144 * 350
d = collections.defaultdict(dict)
for f in glob.glob(os.path.join("gpurun_out", "prof_" + tag, "pmc_*_per_kernel.csv")):
    for r in csv.DictReader(open(f)):
        d[r["kernel"]][r["counter"]] = float(r["per_dispatch"])
for k, v in sorted(d.items()):
    if "SQ_WAVE_CYCLES" not in v: continue
    wc = v["SQ_WAVE_CYCLES"]
    g = lambda n: v.get(n, 0.0)
    print(k)
    print("  waves %d active_any %.1f%% active_valu %.1f%% wait_any %.1f%% wait_inst %.1f%% | VALU busy/SIMD-cycles est: valu_active*4/1024 = %.3g cyc" % (
        g("SQ_WAVES"), 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_ACTIVE_INST_VALU") / wc, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc,
        g("SQ_ACTIVE_INST_VALU") * 4 / 1024))
    print("  per residue: valu wave-insts %.2f (thread util %.0f%%) vmem_rd %.3f vmem_wr %.3f lds %.3f salu %.2f | fetch %.1f B (x2 corrected) write %.1f B" % (
        g("SQ_INSTS_VALU") / R, 100 * g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_ACTIVE_INST_VALU") * 64, 1), g("SQ_INSTS_VMEM_RD") / R, g("SQ_INSTS_VMEM_WR") / R,
        g("SQ_INSTS_LDS") / R, g("SQ_INSTS_SALU") / R, 2 * g("FETCH_SIZE") * 1024 / R, g("WRITE_SIZE") * 1024 / R))
    if "TCC_EA0_RDREQ_DRAM_32B_sum" in v:
        print("  per residue, 32-byte request pieces towards DRAM: read %.1f B write %.1f B" % (32.0 * g("TCC_EA0_RDREQ_DRAM_32B_sum") / R, 32.0 * g("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum") / R))

# calibration (tools/pmc_calibrate.hip: every kernel reads 2^30 and writes 2^30 bytes, last two slightly less)
cal = {}
known = {"fczcal_copy16": 1 << 30, "fczcal_copy4": 1 << 30, "fczcal_lane_stream8": ((1 << 30) // 2800) * 2800, "fczcal_stride100": ((1 << 30) // 100) * 100}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join("gpurun_out", "prof_" + tag, "cal_%s.csv" % ctr)
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            k = r["kernel"].replace("void ", "")
            if k in known: cal[(k, ctr)] = float(r["per_dispatch"]) * 1024 / known[k]
if cal:
    print("calibration: counter KiB*1024 / known bytes")
    for (k, c), v in sorted(cal.items()): print("  %-22s %-10s %.3f" % (k, c, v))
# traffic.json for bench.py. Counter scale from the calibration: every fully-reused pattern (16 B and 4 B coalesced,
# 100-byte lane stride) reads back FETCH_SIZE = 0.50 x bytes and WRITE_SIZE = 1.00 x bytes, so FETCH is doubled and
# WRITE taken as is. The lane-streaming pattern (k_backbone) shows 1.73x / 1.61x on top of that: real sector
# over-fetch / partial-line writes of that pattern, which must stay visible in the traffic figure.
import json
out = {"source": "profiles/%s_pmc_per_kernel.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, %d residues per launch; "
                 "FETCH x2, WRITE x1 per profiles/%s_pmc_summary.txt calibration)" % (tag, int(R), tag),
       "residues_per_chain": 350, "calibration": {"%s %s" % k: round(v, 3) for k, v in sorted(cal.items())}, "kernels": {}}
def short(k):
    """fcz::k_name / void fcz::k_name<args> -> k_name (the exact-mode instantiations), k_name_fast for <true>"""
    n = k.replace("void ", "").split("::")[-1]
    base, _, arg = n.partition("<")
    arg = arg.rstrip(">")
    return base if arg in ("", "0", "false") else (base + "_fast" if arg == "true" else base + "_" + arg)
for k, v in d.items():
    if "fcz::k_" not in k or "FETCH_SIZE" not in v: continue
    out["kernels"][short(k)] = {"fetch_bytes_per_residue": round(2 * v["FETCH_SIZE"] * 1024 / R, 2),
                                        "write_bytes_per_residue": round(v.get("WRITE_SIZE", 0.0) * 1024 / R, 2)}
    if "TCC_EA0_RDREQ_DRAM_32B_sum" in v and "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" in v:
        # exact units (32-byte pieces of the L2's requests towards the memory controllers): preferred by bench.py when present
        out["kernels"][short(k)]["dram_read_bytes_per_residue"] = round(32.0 * v["TCC_EA0_RDREQ_DRAM_32B_sum"] / R, 2)
        out["kernels"][short(k)]["dram_write_bytes_per_residue"] = round(32.0 * v["TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"] / R, 2)
    if "SQ_WAVE_CYCLES" in v:
        # the other bound: VALU issue. Share of its lifetime a wavefront spends issuing VALU instructions, and the
        # VALU wave-instructions per residue (x resident wavefronts per SIMD = share of the SIMD's cycles that issue VALU work)
        out["kernels"][short(k)]["valu_active_share_of_wave_cycles"] = round(v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_WAVE_CYCLES"], 4)
        out["kernels"][short(k)]["valu_wave_insts_per_residue"] = round(v.get("SQ_INSTS_VALU", 0.0) / R, 3)
if out["kernels"]:
    # the kernels these passes describe are the ones compiled from the tree as it stands (the profile runs the tree's own .so):
    # bench.py only quotes the figures while the hash of the kernel sources is this one
    sys.path.insert(0, os.getcwd())
    try:
        import bench
        out["csrc_sha16"] = bench.csrc_sha16(); out["kernel_sources"] = list(bench.KERNEL_SOURCES)
    except Exception as e:   # (bench.py does not import where the summary is made: no hash, and bench.py will not quote the figures)
        print("warning: bench.csrc_sha16 unavailable (%s): traffic.json carries no source hash" % e, file=sys.stderr)
    out["kernel_set"] = sorted(out["kernels"])
    json.dump(out, open(os.path.join("gpurun_out", "prof_" + tag, "traffic.json"), "w"), indent=1)
