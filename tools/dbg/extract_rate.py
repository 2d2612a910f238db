#!/usr/bin/env python3
"""extract / check / decompress-to-directory rates of the C++ host beside the reference's own command line on one database of
synthetic chains (a measurement aid, not a test): python tools/dbg/extract_rate.py [--chains 200000] [--residues 350]"""
import argparse, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=200000)
    ap.add_argument("--residues", type=int, default=350)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--repeats", type=int, default=3, help="runs per command; the fastest is reported")
    ap.add_argument("--decompress-entries", type=int, default=0, help="also: decompress the first N entries into a directory and into a database, both hosts")
    a = ap.parse_args()
    import numpy as np
    import torch
    torch.cuda.init()
    from foldcomp_amd import synthetic
    from foldcomp_amd.codec import Codec
    from foldcomp_amd.database import DatabaseWriter
    c = Codec(0)
    tmp = tempfile.mkdtemp(prefix="fcz_extract_", dir=os.environ.get("TMPDIR") or "/tmp")
    db = os.path.join(tmp, "db")
    w = DatabaseWriter(db)
    key = 0
    for s in range(0, a.chains, 50000):
        n = min(50000, a.chains - s)
        b = synthetic.to_chain_batch(synthetic.generate(n, [a.residues] * n, seed=11 + s, device="cuda"))
        blob, off, st = c.compress_batch(b)
        for i in range(n):
            w.append(blob[off[i]:off[i + 1]].tobytes(), key, f"AF-{key:08d}-F1-model_v4"); key += 1
    w.close()
    c.close()
    host, ref = os.path.join(ROOT, "host", "foldcomp"), os.path.join(ROOT, "oracle", "_ref", "foldcomp_ref")
    out = {"chains": a.chains, "residues": a.residues, "db_bytes": os.path.getsize(db)}
    for tag, args in (("extract_fasta", ["extract", "--fasta"]), ("extract_plddt", ["extract", "--plddt"]), ("extract_plddt_p3", ["extract", "--plddt", "-p", "3"]), ("check", ["check"])):
        row = {}
        for who, exe in (("host", host), ("reference", ref)):
            if not os.path.exists(exe):
                continue
            o = os.path.join(tmp, f"{tag}_{who}.out")
            cmd = [exe, *args, "-t", str(a.threads), db] + ([o] if args[0] == "extract" else [])
            if who == "host" and args[0] == "extract":
                cmd.insert(2, "--json-stats")
            best = None
            for rep in range(a.repeats):
                t0 = time.perf_counter()
                r = subprocess.run(cmd, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, r)
            dt, r = best
            row[who] = {"wall_s": round(dt, 3), "entries_per_s": round(a.chains / dt), "rc": r.returncode, "out_bytes": os.path.getsize(o) if os.path.exists(o) else None}
            if who == "host" and args[0] == "extract":
                try:
                    row[who]["stats"] = json.loads(r.stdout.strip().splitlines()[-1])
                except Exception:
                    row[who]["stats"] = r.stdout[-300:] + r.stderr[-300:]
        if "host" in row and "reference" in row and args[0] == "extract":
            la = sorted(open(os.path.join(tmp, f"{tag}_host.out"), "rb").read().split(b">" if tag != "extract_plddt_p3" else b"\n"))
            lb = sorted(open(os.path.join(tmp, f"{tag}_reference.out"), "rb").read().split(b">" if tag != "extract_plddt_p3" else b"\n"))
            row["same_records"] = la == lb
        out[tag] = row
    if a.decompress_entries:
        # decompress into a directory (the reference's default output: one .pdb per entry) and into a database, the first N entries
        ids = os.path.join(tmp, "ids.txt")
        with open(ids, "w") as fh:
            fh.write("".join(f"AF-{k:08d}-F1-model_v4\n" for k in range(min(a.decompress_entries, a.chains))))
        n = min(a.decompress_entries, a.chains)
        for tag, extra in (("decompress_dir", []), ("decompress_db", ["-d"])):
            row = {}
            for who, exe in (("host", host), ("reference", ref)):
                if not os.path.exists(exe):
                    continue
                o = os.path.join(tmp, f"{tag}_{who}")
                cmd = [exe, "decompress", "-y", "-t", str(a.threads), "--id-list", ids, *extra, db, o]
                if who == "host":
                    cmd.insert(2, "--json-stats")
                t0 = time.perf_counter()
                r = subprocess.run(cmd, capture_output=True, text=True)
                dt = time.perf_counter() - t0
                if os.path.isdir(o):
                    nb = sum(os.path.getsize(os.path.join(o, f)) for f in os.listdir(o)); nf = len(os.listdir(o))
                else:
                    nb = os.path.getsize(o) if os.path.exists(o) else None; nf = 1
                row[who] = {"wall_s": round(dt, 3), "entries_per_s": round(n / dt), "residues_per_s": round(n * a.residues / dt), "rc": r.returncode, "out_bytes": nb, "out_files": nf,
                            "text_GB_per_s": round((nb or 0) / dt / 1e9, 2)}
                if who == "host":
                    try:
                        row[who]["stats"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                    except Exception:
                        row[who]["stats"] = r.stdout[-300:] + r.stderr[-300:]
            if tag == "decompress_dir" and "host" in row and "reference" in row:
                da, dbb = os.path.join(tmp, f"{tag}_host"), os.path.join(tmp, f"{tag}_reference")
                names = sorted(os.listdir(da))
                row["same_files"] = names == sorted(os.listdir(dbb)) and all(open(os.path.join(da, f), "rb").read() == open(os.path.join(dbb, f), "rb").read() for f in names[::max(1, len(names) // 500)])
            out[tag] = row
            import shutil
            for who in ("host", "reference"):
                o = os.path.join(tmp, f"{tag}_{who}")
                if os.path.isdir(o):
                    shutil.rmtree(o, ignore_errors=True)
                else:
                    for ext in ("", ".index", ".lookup", ".dbtype"):
                        if os.path.exists(o + ext):
                            os.remove(o + ext)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
