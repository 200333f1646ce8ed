"""Preprocessed-dataset loader (host mirror of reference code/dataset.py) + synthetic
Avazu/Criteo-shaped data (no real data is reachable offline).

On-disk contract (produced by the reference's data_preprocess/*.py):
  <data_dir>/<name>-meta.json   field_names, feat_map (str -> global id), field_map
  <data_dir>/<name>.h5          feat_ids int64 [N,F], labels int64 [N]   (h5py, or mapx/h5lite.py = libhdf5
                                through ctypes when h5py is not installed), or
  <data_dir>/<name>.npz         the same two arrays (np.savez)
  <data_dir>/split.pkl          {train,valid,test}_index
  <data_dir>/feat-count.pt      float32 [V] train-split id counts (built on first pretrain run)
"""
import json
import os
import pickle as pkl

import numpy as np
import torch

SPLITS = ("train", "valid", "test")


class OurDataset(torch.utils.data.Dataset):
    def __init__(self, X, Y):
        self.X, self.Y = X, Y

    def __len__(self):
        return len(self.Y)

    def __getitem__(self, k):
        return self.X[k], self.Y[k]


def _read_table(data_dir, name):
    npz = os.path.join(data_dir, f"{name}.npz")
    h5 = os.path.join(data_dir, f"{name}.h5")
    if os.path.exists(npz):
        z = np.load(npz)
        return z["feat_ids"], z["labels"]
    if os.path.exists(h5):
        try:
            import h5py
        except ImportError:
            from . import h5lite                      # libhdf5 through ctypes (h5py is not in the ROCm image)
            t = h5lite.read_datasets(h5, ["feat_ids", "labels"])
            return t["feat_ids"], t["labels"]
        with h5py.File(h5, "r") as f:
            return f["feat_ids"][:], f["labels"][:]
    raise FileNotFoundError(f"neither {npz} nor {h5} exists")


class BaseDataset:
    def __init__(self, args):
        self.args = args
        self.data_dir, self.dataset_name = args.data_dir, args.dataset_name
        self.split_names = list(SPLITS)
        self.load_data()

    def load_data(self):
        with open(os.path.join(self.data_dir, f"{self.dataset_name}-meta.json"), "r") as f:
            meta = json.load(f)
        self.field_names, self.feat_map, self.field_map = meta["field_names"], meta["feat_map"], meta["field_map"]
        feat_ids, labels = _read_table(self.data_dir, self.dataset_name)
        with open(os.path.join(self.data_dir, "split.pkl"), "rb") as f:
            split_index = pkl.load(f)
        self.X = {s: feat_ids[split_index[f"{s}_index"]] for s in self.split_names}
        self.Y = {s: labels[split_index[f"{s}_index"]] for s in self.split_names}
        self.get_feat_count_file()
        self.count_feat_per_field(feat_ids)

    def get_splited_dataset(self, split):
        assert split in self.split_names, f"Unsupported split name: {split}"
        return OurDataset(self.X[split], self.Y[split])

    def get_feat_count_file(self):
        """feat-count.pt: occurrences of every global id in the TRAIN split (dataset.py:49-60;
        the reference counts with a Python Counter over N*F ints — np.bincount here)."""
        path = os.path.join(self.data_dir, "feat-count.pt")
        if not self.args.pretrain:
            self.feat_count = None
        elif os.path.exists(path):
            self.feat_count = torch.load(path)
        else:
            # every rank counts for itself (np.bincount, deterministic); rank 0 alone writes the
            # file, atomically, so no rank ever reads a half-written one
            cnt = np.bincount(self.X["train"].reshape(-1), minlength=len(self.feat_map))
            self.feat_count = torch.from_numpy(cnt.astype(np.float32))
            if int(os.environ.get("RANK", "0")) == 0:
                tmp = f"{path}.tmp.{os.getpid()}"
                torch.save(self.feat_count, tmp)
                os.replace(tmp, path)

    def count_feat_per_field(self, feat_ids):
        if self.args.pt_type == "RFD" and self.args.RFD_replace == "Uniform":
            self.idx_low = torch.from_numpy(feat_ids.min(axis=0))
            self.idx_high = torch.from_numpy(feat_ids.max(axis=0) + 1)
            self.feat_num_per_field = self.idx_high - self.idx_low
        else:
            self.idx_low = self.idx_high = self.feat_num_per_field = None


# ------------------------------------------------------------------------------- synthetic
AVAZU_F23_V = 9449445       # 9 449 435 feature values + 10 reserved ids
CRITEO_F39_V = 33762577


def field_sizes(num_fields, vocab, reserved=10):
    """Avazu-like cardinality skew: field sizes geometrically spaced from 2 values up to a
    top size chosen (bisection) so that they sum to vocab - reserved; F=23, V=9 449 445 gives
    {2, 4, 8, 15, ... 1.2 M, 2.4 M, 4.7 M}."""
    target = vocab - reserved
    assert target >= 2 * num_fields, "vocab too small"
    lo, hi = 2.0, float(target)
    for _ in range(200):
        top = 0.5 * (lo + hi)
        sizes = np.maximum(2, np.round(np.geomspace(2.0, top, num_fields))).astype(np.int64)
        if sizes.sum() > target:
            hi = top
        else:
            lo = top
    sizes = np.maximum(2, np.round(np.geomspace(2.0, lo, num_fields))).astype(np.int64)
    sizes[-1] += target - sizes.sum()
    assert sizes.min() >= 2 and sizes.sum() == target
    return sizes


def _zipf_ranks(rng, size, n, s=1.1):
    """Zipf(s)-distributed ranks in [0, n): inverse-CDF on the continuous approximation."""
    u = rng.random(size)
    if n == 1:
        return np.zeros(size, dtype=np.int64)
    a = 1.0 - s
    x = ((u * ((n + 1.0) ** a - 1.0)) + 1.0) ** (1.0 / a) - 1.0
    return np.minimum(x.astype(np.int64), n - 1)


def synth_table(num_rows, num_fields=23, vocab=AVAZU_F23_V, seed=42, zipf=1.1, uniform=False):
    """-> (feat_ids int64 [N,F], labels int64 [N], field_low int64 [F], field_high int64 [F]).
    Field f owns a contiguous id range; inside a field rank 0 (lowest id) is the most
    frequent value, matching the most_common() ordering of proc_avazu.py:248-251."""
    rng = np.random.default_rng(seed)
    sizes = field_sizes(num_fields, vocab)
    low = 10 + np.concatenate([[0], np.cumsum(sizes)[:-1]])
    ids = np.empty((num_rows, num_fields), dtype=np.int64)
    for f in range(num_fields):
        r = rng.integers(0, sizes[f], num_rows) if uniform else _zipf_ranks(rng, num_rows, int(sizes[f]), zipf)
        ids[:, f] = low[f] + r
    labels = (rng.random(num_rows) < 0.17).astype(np.int64)
    return ids, labels, low, low + sizes


def write_synth_dataset(data_dir, name="avazu", num_rows=10000, num_fields=23, vocab=2000, seed=42):
    """A small on-disk dataset in the reference's layout (npz table) for run.py tests."""
    os.makedirs(data_dir, exist_ok=True)
    ids, labels, low, high = synth_table(num_rows, num_fields, vocab, seed)
    field_names = [f"C{f}" for f in range(num_fields)]
    feat_map = {t: i for i, t in enumerate(["<pad>", "<cls>", "<sep>", "<mask>"] + [f"<unused{i}>" for i in range(6)])}
    for f in range(num_fields):
        for v in range(int(high[f] - low[f])):
            feat_map[f"{field_names[f]}-{v}"] = int(low[f]) + v
    field_map = {"<rsv>": 0, **{n: i + 1 for i, n in enumerate(field_names)}}
    with open(os.path.join(data_dir, f"{name}-meta.json"), "w") as f:
        json.dump(dict(field_names=field_names, feat_map=feat_map, field_map=field_map), f)
    np.savez(os.path.join(data_dir, f"{name}.npz"), feat_ids=ids, labels=labels)
    perm = np.random.default_rng(seed + 1).permutation(num_rows)
    a, b = int(num_rows * 0.8), int(num_rows * 0.9)
    with open(os.path.join(data_dir, "split.pkl"), "wb") as f:
        pkl.dump(dict(train_index=np.sort(perm[:a]), valid_index=np.sort(perm[a:b]),
                      test_index=np.sort(perm[b:])), f)
    return data_dir
