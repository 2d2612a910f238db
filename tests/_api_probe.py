"""One fixed sequence of calls against whatever module `foldcomp` is on the path -- the reference's own extension (oracle/_ref/pymod,
built by oracle/build_ref.sh) or the drop-in (foldcomp/ -> foldcomp_amd) -- with every result and every exception written as JSON.
tests/test_api_vs_reference_module.py runs it once per module and compares the two documents. Usage: _api_probe.py <fixtures.npz> <workdir>"""
import json
import os
import sys

import numpy as np

import foldcomp


def mask(b):
    a = bytearray(b)
    if a[:4] == b"FCMP":
        for i in (14, 15, 22, 23):
            if i < len(a):
                a[i] = 0
    return bytes(a).hex()


def enc(v):
    if isinstance(v, bytes):
        return {"bytes": mask(v)}
    if isinstance(v, float):
        return {"f": v.hex()}
    if isinstance(v, (tuple, list)):
        return {"seq" if isinstance(v, list) else "tuple": [enc(x) for x in v]}
    if isinstance(v, dict):
        return {"dict": {k: enc(x) for k, x in sorted(v.items())}}
    if v is None or isinstance(v, (bool, int, str)):
        return v
    return {"type": type(v).__name__}


def cap(f):
    try:
        return ["ok", enc(f())]
    except BaseException as e:   # noqa: BLE001
        return ["raised", "foldcomp.error" if isinstance(e, foldcomp.error) else type(e).__name__, str(e)]


def main():
    z = np.load(sys.argv[1])
    work = sys.argv[2]
    f = {k[5:]: z[k].tobytes() for k in z.keys() if k.startswith("file:")}
    out = {}
    af = f["test_af.pdb"].decode()
    big = f["test.pdb"].decode()
    multi = f["multichain.pdb"].decode()
    # ---- compress
    out["compress"] = cap(lambda: foldcomp.compress("test_af", af))
    out["compress_big"] = cap(lambda: foldcomp.compress("a longer title, with blanks", big))
    for b in (1, 10, 200):
        out[f"compress_b{b}"] = cap(lambda b=b: foldcomp.compress("t", af, anchor_residue_threshold=b))
    out["compress_multi"] = cap(lambda: foldcomp.compress("m", multi))
    out["compress_chains"] = cap(lambda: [foldcomp.compress("m", s) for s in foldcomp.split_pdb_by_chain(multi)] if hasattr(foldcomp, "split_pdb_by_chain") else "no split_pdb_by_chain")
    out["compress_empty"] = cap(lambda: foldcomp.compress("e", ""))
    out["compress_no_atoms"] = cap(lambda: foldcomp.compress("e", "HEADER only\nEND\n"))
    out["compress_bytes_arg"] = cap(lambda: foldcomp.compress("e", af.encode()))
    out["compress_bad_threshold"] = cap(lambda: foldcomp.compress("e", af, anchor_residue_threshold="25"))
    out["compress_positional_threshold"] = cap(lambda: foldcomp.compress("e", af, 25))
    # an alternative position: the same atom twice in a row (columns 17: 'A' / 'B'); removeAlternativePosition keeps the first
    ls = af.splitlines(keepends=True)
    k = next(i for i, l in enumerate(ls) if l.startswith("ATOM") and l[12:16].strip() == "CB")
    alt = ls[:k] + [ls[k][:16] + "A" + ls[k][17:], ls[k][:16] + "B" + ls[k][17:30] + "   1.000   2.000   3.000" + ls[k][54:]] + ls[k + 1:]
    out["compress_altloc"] = cap(lambda: foldcomp.compress("h", "".join(alt)))
    out["get_data_altloc"] = cap(lambda: foldcomp.get_data("".join(alt)))
    # a backbone atom that is a HETATM record: the module reads ATOM lines only, the residue loses its N
    out["compress_missing_backbone_atom"] = cap(lambda: foldcomp.compress("h", af.replace("ATOM      9", "HETATM    9", 1)))
    fcz = foldcomp.compress("test_af", af)
    # ---- decompress
    out["decompress"] = cap(lambda: foldcomp.decompress(fcz))
    out["decompress_big"] = cap(lambda: foldcomp.decompress(foldcomp.compress("x", big)))
    out["decompress_garbage"] = cap(lambda: foldcomp.decompress(b"not an fcz record at all"))
    out["decompress_empty"] = cap(lambda: foldcomp.decompress(b""))
    out["decompress_str_arg"] = cap(lambda: foldcomp.decompress("text"))
    out["decompress_trailing_bytes"] = cap(lambda: foldcomp.decompress(fcz + b"\0\0\0"))
    # ---- get_data
    out["get_data_fcz"] = cap(lambda: foldcomp.get_data(fcz))
    out["get_data_pdb"] = cap(lambda: foldcomp.get_data(af))
    out["get_data_garbage_bytes"] = cap(lambda: foldcomp.get_data(b"zzzz"))
    out["get_data_int"] = cap(lambda: foldcomp.get_data(5))
    # ---- open: the reference's own example database (MMseqs layout: every entry ends in a NUL)
    db = os.path.join(work, "example_db")
    for ext in ("", ".index", ".lookup", ".dbtype"):
        with open(db + ext, "wb") as fh:
            fh.write(f["example_db" + ext])
    names = [l.split("\t")[1] for l in f["example_db.lookup"].decode().splitlines()]

    def walk(**kw):
        with foldcomp.open(db, **kw) as d:
            return [len(d), [x for x in d]]
    out["open_all"] = cap(lambda: walk())
    out["open_raw"] = cap(lambda: walk(decompress=False))
    out["open_ids"] = cap(lambda: walk(ids=[names[5], names[0], names[5]]))
    out["open_ids_missing"] = cap(lambda: walk(ids=[names[1], "absent", names[2]]))
    out["open_ids_missing_err"] = cap(lambda: walk(ids=[names[1], "absent"], err_on_missing=True))
    out["open_empty_ids"] = cap(lambda: walk(ids=[]))
    out["open_ids_not_list"] = cap(lambda: walk(ids=(names[0],)))
    out["open_decompress_not_bool"] = cap(lambda: walk(decompress=1))
    out["open_err_not_bool"] = cap(lambda: walk(err_on_missing="yes"))
    out["open_positional_ids"] = cap(lambda: foldcomp.open(db, [names[0]]))
    out["open_pathlike"] = cap(lambda: len(foldcomp.open(__import__("pathlib").Path(db))))
    d = foldcomp.open(db)
    out["index_last"] = cap(lambda: d[len(d) - 1])
    out["index_past_end"] = cap(lambda: d[len(d)])
    out["index_far"] = cap(lambda: d[10 ** 6])
    out["index_negative_one"] = cap(lambda: d[-1])                       # the sequence protocol adds len() to a negative index
    out["len"] = cap(lambda: len(d))
    out["random_access"] = cap(lambda: [d[7][0], d[2][0], d[3][0], d[3][0], d[0][0]])
    out["close"] = cap(lambda: d.close())
    out["close_twice"] = cap(lambda: d.close())
    # ---- extra inputs (tests/test_api_vs_reference_module.py: the differential fuzz's variants rendered as PDB text, composite files):
    #      compress, decompress of the result, get_data of the text and of the record
    if len(sys.argv) > 3:
        x = np.load(sys.argv[3])
        forked = len(sys.argv) > 4 and sys.argv[4] == "fork"      # (the reference's module ends the PROCESS on some texts -- std::stof throws
        for k in sorted(x.keys()):                                #  through a noexcept frame --: each input in a child of its own)
            text = x[k].tobytes().decode("latin-1")

            def one():
                res = {}
                rec = cap(lambda: foldcomp.compress(k, text))
                res["x_compress:" + k] = rec
                res["x_get_data_text:" + k] = cap(lambda: foldcomp.get_data(text))
                if rec[0] == "ok":
                    raw = foldcomp.compress(k, text)
                    res["x_decompress:" + k] = cap(lambda: foldcomp.decompress(raw))
                    res["x_get_data_fcz:" + k] = cap(lambda: foldcomp.get_data(raw))
                return res
            if not forked:
                out.update(one())
                continue
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:
                os.close(r)
                try:
                    with os.fdopen(w, "w") as fh:
                        json.dump(one(), fh)
                finally:
                    os._exit(0)
            os.close(w)
            with os.fdopen(r) as fh:
                data = fh.read()
            _, status = os.waitpid(pid, 0)
            try:
                out.update(json.loads(data) if status == 0 else {"x_compress:" + k: ["process ended", status]})
            except ValueError:
                out["x_compress:" + k] = ["process ended", status]
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()
