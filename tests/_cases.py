"""Shared helpers: golden cases -> ChainBatch, seeded synthetic batches."""
import numpy as np

from foldcomp_amd.structure import ChainBatch

_KEYS = ("res_off", "atom_off", "x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index", "first_atom_index",
         "chain_id", "titles", "title_off")


def golden_batch(z, name) -> ChainBatch:
    kw = {k: np.ascontiguousarray(z[f"{name}/in/{k}"]) for k in _KEYS}
    return ChainBatch(anchor_threshold=int(z[f"{name}/in/anchor_threshold"][0]), **kw)


def compress_cases(index):
    return [n for n in index if n.startswith("pdb:") or n.startswith("syn:")]


def db_cases(index):
    return [n for n in index if n.startswith("db:")]


def concat_batches(batches):
    """several ChainBatch -> one (same anchor threshold)"""
    res_off = [0]; atom_off = []; abase = 0; toff = [0]
    cat = {k: [] for k in ("x", "y", "z", "atom_code", "res_code", "bfac_ca", "first_res_index", "first_atom_index", "chain_id", "titles")}
    for b in batches:
        res_off += list(res_off[-1] + b.res_off[1:].astype(np.int64))
        atom_off.append(b.atom_off[:-1].astype(np.int64) + abase); abase += int(b.atom_off[-1])
        toff += list(toff[-1] + b.title_off[1:].astype(np.int64))
        for k in cat: cat[k].append(getattr(b, k))
    atom_off.append(np.asarray([abase]))
    return ChainBatch(res_off=np.asarray(res_off, np.uint32), atom_off=np.concatenate(atom_off).astype(np.uint32),
                      title_off=np.asarray(toff, np.uint32), anchor_threshold=batches[0].anchor_threshold,
                      **{k: np.concatenate(v) for k, v in cat.items()})


def entries_blob(entries):
    """list of bytes -> (blob, off)"""
    off = np.zeros(len(entries) + 1, np.uint64)
    off[1:] = np.cumsum([len(e) for e in entries])
    blob = np.frombuffer(b"".join(entries), np.uint8).copy() if entries else np.zeros(0, np.uint8)
    return blob, off
