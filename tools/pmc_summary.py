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
