#!/usr/bin/env python3
"""A stand-in for `host/foldcomp-hip <mode> -d --shard R/N [--place] ... <input> <output>` that needs no GPU: it speaks the engine's side
of the sharded drivers' protocol (foldcomp_amd/sharded_cli.py) with made-up records, so that the CPU suite can hold the PYTHON side of
that protocol -- counts first / placement / one write for decompress, partial database + splice for compress, failures at every stage
-- to what a single writer produces. Test infrastructure only.

The "input" is a text file of lines `<name> <length>`; rank R of N takes the lines i with i % N == R ... no: a contiguous range, like
the real engine (cut by cumulative length with shard.shard_cuts). A record's bytes are its name repeated to its length.
FAIL_AT in the environment: "sizes:<rank>" / "place:<rank>" / "write:<rank>" makes that rank fail at that stage."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def record(name, n):
    return (name.encode() * (n // max(len(name), 1) + 1))[:n]


def main():
    argv = sys.argv[1:]
    mode = argv[0]
    rank, world = 0, 1
    place = "--place" in argv
    for i, a in enumerate(argv):
        if a == "--shard":
            rank, world = (int(x) for x in argv[i + 1].split("/"))
    inp, out = argv[-2], argv[-1]
    fail = os.environ.get("FAIL_AT", "")
    items = [(l.split()[0], int(l.split()[1])) for l in open(inp).read().splitlines() if l.strip()]
    from foldcomp_amd.shard import shard_cuts
    cuts = shard_cuts([n for _, n in items], world)
    mine = items[cuts[rank]:cuts[rank + 1]]
    nbytes = sum(n for _, n in mine)
    key0 = off0 = 0
    if place:
        if fail == f"sizes:{rank}":
            print(json.dumps({"phase": "sizes", "records": 0, "data_bytes": 0, "failed": True})); sys.stdout.flush(); sys.exit(1)
        print(json.dumps({"phase": "sizes", "records": len(mine), "data_bytes": nbytes, "failed": False, "sizes_pass_s": 0.01})); sys.stdout.flush()
        line = sys.stdin.readline()
        parts = line.split()
        if len(parts) != 3 or fail == f"place:{rank}":
            sys.exit(1)
        key0, off0, total = int(parts[0]), int(parts[1]), int(parts[2])
    if fail == f"write:{rank}":
        sys.exit(1)
    tag = f".{rank}" if (place and rank > 0) else ""
    fd = os.open(out, os.O_CREAT | os.O_WRONLY | (0 if place else os.O_TRUNC), 0o666)
    with open(out + ".index" + tag, "w") as fi, open(out + ".lookup" + tag, "w") as fl:
        o = off0
        for k, (name, n) in enumerate(mine):
            os.pwrite(fd, record(name, n), o)
            fi.write(f"{key0 + k}\t{o}\t{n}\n"); fl.write(f"{key0 + k}\t{name}\t0\n"); o += n
    if place and rank == 0:
        os.ftruncate(fd, total)
    os.close(fd)
    if not place or rank == 0:
        open(out + ".dbtype", "wb").write((12).to_bytes(4, "little"))
    print(json.dumps({"mode": mode, "records": len(mine), "data_bytes": nbytes, "residues": 10 * len(mine), "wall_s": 0.05, "ctx_ready_s": 0.01,
                      "max_rss_kb": 1000, "items": len(mine), "fcz_bytes": nbytes, "sizes_pass_s": 0.01 if place else 0.0, "placed": place}))


if __name__ == "__main__":
    main()
