#!/usr/bin/env python3
"""Builds profiles/traffic.json from two rocprofv3 PMC passes (separate runs, as MI355X_MICROARCH.md prescribes):
    tools/pmc_run.sh fetch FETCH_SIZE ; tools/pmc_run.sh write WRITE_SIZE
    python tools/traffic_json.py gpurun_out/pmc_fetch/pmc_results.db gpurun_out/pmc_write/pmc_results.db profiles/traffic.json
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced read stream (guide, HBM section); Infinity-Cache hits are counted."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    out = {}
    for k, v, n in rows:
        name = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if name.startswith("at::") or name.startswith("__amd"):
            continue
        out[name] = (v / n, n)
    return out


def main(fetch_db, write_db, dst):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {"note": __doc__.split("\n", 4)[4].strip().replace("\n", " "), "per_kernel": {}}
    for k in sorted(f):
        fk, n = f[k]
        wk = w.get(k, (0.0, 0))[0]
        res["per_kernel"][k] = {"FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1), "dispatches": n,
                                "bytes_per_launch": int((2 * fk + wk) * 1024)}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res["per_kernel"], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
