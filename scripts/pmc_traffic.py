#!/usr/bin/env python3
"""HBM bytes per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes).
Usage: pmc_traffic.py FETCH.csv WRITE.csv KERNEL_SUBSTRING OUT.json "command" COMMIT GAMES ROUND SIMS ITERS"""
import csv
import json
import sys


def per_launch(path, needle, counter):
    vals = []
    with open(path) as f:
        rd = csv.reader(f)
        head = next(rd)
        kn, cn, cv = head.index("Kernel_Name"), head.index("Counter_Name"), head.index("Counter_Value")
        for r in rd:
            if needle in r[kn] and r[cn] == counter:
                vals.append(float(r[cv]))
    return (sum(vals) / len(vals) if vals else None), len(vals)


fetch, nf = per_launch(sys.argv[1], sys.argv[3], "FETCH_SIZE")
write, nw = per_launch(sys.argv[2], sys.argv[3], "WRITE_SIZE")
out = {"FETCH_SIZE_KB_per_launch": fetch, "FETCH_SIZE_dispatches": nf, "WRITE_SIZE_KB_per_launch": write,
       "WRITE_SIZE_dispatches": nw, "command": sys.argv[5], "kernel": sys.argv[3], "commit": sys.argv[6],
       "config": [int(v) for v in sys.argv[7:11]],
       "hbm_bytes_per_launch_raw": (fetch + write) * 1024.0,
       "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0,
       "note": "FETCH_SIZE / WRITE_SIZE are in KB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly half the bytes of "
               "16-B-per-lane reads -- the tree-node reads of this kernel are 16 B per lane (three per 48-byte node) and its "
               "weight stream is 16 B per lane, so `hbm_bytes_per_launch` = 2 x FETCH_SIZE + WRITE_SIZE; the uncorrected sum "
               "is `hbm_bytes_per_launch_raw`.  Infinity-Cache hits are counted too (memory-side requests of the L2)."}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
