#!/usr/bin/env python3
"""Average FETCH_SIZE / WRITE_SIZE per launch per kernel from two rocprofv3 --pmc passes.
Units: the counters are in KiB (hbm_bytes = value * 1024, cdna_hip_programming.md §7);
gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streams
(x2; MI355X_MICROARCH.md §HBM) — reported both raw and x2.
    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return {k: (v[0] / v[1] * 1024.0, v[1]) for k, v in acc.items()}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("mapx::"):
        continue
    f, n = fetch.get(k, (0.0, 0))
    w, _ = write.get(k, (0.0, 0))
    out[k] = dict(launches=n, fetch_bytes_raw=f, fetch_bytes_x2=2 * f, write_bytes=w,
                  hbm_bytes_per_launch=2 * f + w)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
    print(f"{k[:64]:64s} x{v['launches']:4d}  fetch(raw) {v['fetch_bytes_raw'] / 1e6:9.2f} MB  "
          f"fetch(x2) {v['fetch_bytes_x2'] / 1e6:9.2f} MB  write {v['write_bytes'] / 1e6:9.2f} MB")
if len(sys.argv) > 3:
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    out["_meta"] = dict(csrc_hash=h.hexdigest()[:16],        # == bench.csrc_hash(): bench.py flags a mismatch
                        note="HBM bytes per launch; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)")
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
