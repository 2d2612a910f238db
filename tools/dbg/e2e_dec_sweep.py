"""where the time of `foldcomp-hip decompress -d` goes: the same database list through the host with different writer
thread counts, workers per GPU and output file systems (run on the GPU box: python tools/dbg/e2e_dec_sweep.py)"""
import json, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ing = np.load(os.path.join(ROOT, "tests", "golden", "reference_ingest.npz"))
src = tempfile.mkdtemp(prefix="fcz_sweep_", dir="/dev/shm")
for suffix in ("", ".index", ".lookup", ".dbtype"):
    open(os.path.join(src, "example_db" + suffix), "wb").write(ing[f"file:example_db{suffix}"].tobytes())
lst = os.path.join(src, "dbs.txt")
open(lst, "w").write((os.path.join(src, "example_db") + "\n") * int(sys.argv[1] if len(sys.argv) > 1 else 1500))
exe = os.path.join(ROOT, "host", "foldcomp-hip")
for outdir in ("/tmp", "/dev/shm"):
    for t, wpg in ((8, 2), (16, 2), (32, 2), (64, 2), (32, 3), (32, 4), (64, 4)):
        out = os.path.join(outdir, "fcz_sweep_out")
        t0 = time.time()
        r = subprocess.run([exe, "decompress", "-d", "-y", "-t", str(t), "--gpus", "1", "--workers-per-gpu", str(wpg), "--json-stats", "-f", lst, out],
                           capture_output=True, text=True)
        wall = time.time() - t0
        try:
            st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            print(f"{outdir:9s} -t {t:2d} workers/gpu {wpg}: wall {st['wall_s']:.3f} s  (process {wall:.3f})  ctx {st['ctx_ready_s']:.2f}  queued {st['all_queued_s']:.2f}  codec-sum {st['codec_call_s_sum']:.2f}  "
                  f"{st['residues_per_s'] / 1e6:.2f} M res/s  text {st['text_MB_per_s'] / 1e3:.2f} GB/s  records {st['records']}", flush=True)
        except Exception as e:
            print("failed", outdir, t, wpg, r.returncode, r.stderr[-300:])
        for suffix in ("", ".index", ".lookup", ".dbtype"):
            try: os.remove(out + suffix)
            except OSError: pass
shutil.rmtree(src, ignore_errors=True)
