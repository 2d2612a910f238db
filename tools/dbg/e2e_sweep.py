import os, sys, subprocess, time, json, ctypes, shutil, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _harness as H
z = np.load(os.path.join(ROOT, "tests/golden/reference_ingest.npz"))
data = z["file:test.pdb"].tobytes()
d = "/tmp/e2e_sweep"; shutil.rmtree(d, ignore_errors=True); os.makedirs(d + "/pdb")
N = 4096
paths = []
for i in range(N):
    p = f"{d}/pdb/s{i:05d}.pdb"; open(p, "wb").write(data); paths.append(p)
host = os.path.join(ROOT, "host/foldcomp-hip")
for t in (8, 16, 32, 64, 128, 256):
    env = dict(os.environ, OMP_NUM_THREADS=str(t))
    r = subprocess.run([host, "parse-bench", d + "/pdb", "x"], capture_output=True, text=True, env=env)
    print("parse-bench", t, r.stdout.strip())
    r = subprocess.run([host, "compress", "-d", "-y", "--json-stats", d + "/pdb", d + f"/db{t}"], capture_output=True, text=True, env=env)
    st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("compress -d", t, {k: st[k] for k in ("wall_s", "parse_s", "codec_call_s_sum", "residues_per_s")})
rl = H.load_ref()
rl.ref_compress_files.restype = ctypes.c_int
rl.ref_compress_files.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
blob = b"".join(p.encode() + b"\0" for p in paths)
for t in (8, 16, 32, 64, 128, 256):
    secs = ctypes.c_double(); rres = ctypes.c_ulonglong(); rbytes = ctypes.c_ulonglong(); flen = ctypes.c_long(); first = ctypes.create_string_buffer(1 << 20)
    rl.ref_compress_files(blob, N, t, 25, ctypes.byref(secs), ctypes.byref(rres), ctypes.byref(rbytes), first, 1 << 20, ctypes.byref(flen))
    print("reference", t, round(secs.value, 4), "s", round(rres.value / secs.value), "res/s")
shutil.rmtree(d, ignore_errors=True)
