#!/usr/bin/env python3
"""Golden vectors of the vocabulary builders from the REAL reference preprocessing,
data_preprocess/proc_avazu.py and proc_criteo.py generate_dataset(), run on seeded synthetic columns.

Runs only where /root/reference exists (the build container).  What is stored: the synthetic input columns IN
THE ORDER generate_dataset() sees them (after its own seeded shuffle, whose permutation it records in the JSON
it writes) and the ids of its `feat_map` — outputs, no reference source.

Import note: both modules import h5py at the top, which this image does not have (ModuleNotFoundError, an
ordinary Python error).  An EMPTY module object is registered under that name so that the modules import;
generate_dataset() then runs the whole vocabulary construction, writes its JSON (feat_map, index, field_map) and
only afterwards reaches `h5py.File(...)`, where the empty module raises AttributeError — caught here.  Nothing of
HDF5 is emulated; the fixture is read back from the JSON.

    python tests/golden/gen_vocab_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data_preprocess"


def synth_columns(names, n_rows, seed, hexy):
    """Zipf-like columns: a few frequent values, a long tail below any n_core, ties in the counts; integer fields
    and (hexy) 8-digit hexadecimal hash strings like the raw Avazu / Criteo ids."""
    rng = np.random.RandomState(seed)
    cols = {}
    for j, name in enumerate(names):
        vocab = int([2, 7, 30, 200, 1500, 6000][j % 6])
        ranks = np.minimum((rng.pareto(0.9, n_rows) * 2).astype(np.int64), vocab - 1)
        raw = rng.permutation(1 << 20)[:vocab]
        vals = raw[ranks]
        if j % 7 == 3:
            vals = np.where(rng.rand(n_rows) < 0.02, -1, vals)       # the reference's missing-value code
        if name in hexy:
            cols[name] = np.array([format(int(v) & 0xffffffff, "08x") for v in vals])
        else:
            cols[name] = vals.astype(np.int64)
    return cols


def run_reference(module_name, subdir, n_core, n_rows, seed):
    if "h5py" not in sys.modules:
        sys.modules["h5py"] = types.ModuleType("h5py")          # see the docstring
    sys.path.insert(0, REF)
    mod = __import__(module_name)
    names = [n for n in mod.valid_fields if n != "click"]
    hexy = {n for n in names if n.endswith("_id") or n.endswith("_ip") or n.endswith("_domain")
            or n.endswith("_model") or n.endswith("_category") or (n.startswith("C") and module_name == "proc_criteo")}
    cols = synth_columns(names, n_rows, seed, hexy)
    rng = np.random.RandomState(seed + 1)
    click = (rng.rand(n_rows) < 0.17).astype(np.int64)
    with tempfile.TemporaryDirectory() as tmp:
        data_dir = os.path.join(tmp, "data", subdir) + os.sep
        os.makedirs(os.path.join(data_dir, f"{subdir}_x4"))
        work = os.path.join(tmp, "work")
        os.makedirs(work)
        np.save(os.path.join(data_dir, "click.npy"), click)
        for name, c in cols.items():
            np.save(os.path.join(data_dir, f"{name}.npy"), c)
        mod.data_dir = data_dir
        cwd = os.getcwd()
        os.chdir(work)                       # its output paths are '../data/<subdir>/<subdir>_x4/...'
        try:
            mod.generate_dataset(n_core=n_core)
        except AttributeError as e:          # h5py.File on the empty module: the JSON is on disk by now
            assert "File" in str(e), e
        finally:
            os.chdir(cwd)
        meta = json.load(open(os.path.join(data_dir, f"{subdir}_x4", f"{subdir}_x4_{n_core}-core.json")))
    index = np.array(meta["index"])
    feat_map = meta["feat_map"]
    shuffled = {name: c[index] for name, c in cols.items()}
    ids = np.stack([np.array([feat_map.get(f"{name}-{v}", feat_map[f"{name}-<oov>"]) for v in shuffled[name]],
                             dtype=np.int64) for name in names], axis=1)
    out = {"n_core": np.int64(n_core), "names": np.array(names), "feat_ids": ids,
           "input_size": np.int64(len(feat_map)),
           "feat_map_keys": np.array(list(feat_map.keys())), "feat_map_ids": np.array(list(feat_map.values()), dtype=np.int64)}
    for name in names:
        out[f"col/{name}"] = shuffled[name]
    return out


if __name__ == "__main__":
    for module_name, subdir, n_core, n_rows, seed in (("proc_avazu", "avazu", 5, 6000, 11),
                                                     ("proc_criteo", "criteo", 3, 4000, 23)):
        out = run_reference(module_name, subdir, n_core, n_rows, seed)
        path = os.path.join(HERE, f"vocab_{subdir}.npz")
        np.savez_compressed(path, **out)
        print(path, "fields", len(out["names"]), "rows", out["feat_ids"].shape[0], "input_size", int(out["input_size"]),
              os.path.getsize(path) // 1024, "KiB")
