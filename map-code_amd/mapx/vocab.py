"""Vocabulary builder on the GPU: the per-field id assignment of the reference's offline
preprocessing, data_preprocess/proc_avazu.py:213-262 and proc_criteo.py:109-163, for datasets whose raw
columns are resident on the device.

    feat_map: <pad> <cls> <sep> <mask> <unused0..5> = ids 0..9 (proc_avazu.py:213-220), then field by field
    the values seen at least n_core times in Counter.most_common() order — descending count, ties in order of
    first occurrence — and one <oov> id per field (proc_avazu.py:247-250); every row's value is translated
    through it (:252-257).  The resulting matrix is `feat_ids` of the HDF5 file that code/dataset.py:20-40 reads.

Counting / ranking / translating are kernels of csrc/vocab.hip behind include/mapx_hip.h (mapx_vocab_*); the
two stable sorts of a field's distinct values are the segment plans' radix sort (mapx_seg_plan).  Raw values
are 64-bit integers: the hexadecimal hash strings of Avazu / Criteo parse to them exactly (`encode_column`),
anything else is coded by first occurrence on the host.  There is no CPU path."""
import numpy as np
import torch

from . import ops
from .native import MapxError, check, lib, ptr, require_gpu, stream

RESERVED = ("<pad>", "<cls>", "<sep>", "<mask>", "<unused0>", "<unused1>", "<unused2>", "<unused3>", "<unused4>",
            "<unused5>")                                  # proc_avazu.py:213-220: ids 0..9


_HEX = __import__("re").compile(r"[0-9a-f]*")


def encode_column(values):
    """A raw column -> (int64 codes, decode) with decode(code) = the value as the reference's f-string prints
    it (proc_avazu.py:249: f'{name}-{k}').  Integer columns are their own codes; columns of hexadecimal
    strings (the 8-digit hashes of Avazu / Criteo) parse exactly; anything else is numbered by first occurrence."""
    a = np.asarray(values)
    if a.dtype.kind in "iu":
        return a.astype(np.int64), (lambda c: str(int(c)))
    if a.dtype.kind in "USO":
        strs = [str(x) for x in a.tolist()]
        # fixed-width lower-case hexadecimal — EVERY string matches [0-9a-f]{w} (int(s, 16) alone also takes '1A',
        # ' 1f', '1_0', '+1f', which would merge values the reference's Counter keeps apart): code <-> string is a bijection
        w = len(strs[0]) if strs else 0
        if 0 < w <= 15 and all(len(s) == w for s in strs) and _HEX.fullmatch("".join(strs)) is not None:
            codes = np.array([int(s, 16) for s in strs], dtype=np.int64)
            return codes, (lambda c, w=w: format(int(c), f"0{w}x"))
        first = {}
        codes = np.fromiter((first.setdefault(s, len(first)) for s in strs), dtype=np.int64, count=len(strs))
        names = list(first)
        return codes, (lambda c: names[int(c)])
    raise TypeError(f"encode_column: dtype {a.dtype} (integers or strings)")


class FieldVocab:
    """Result for one field: `keys` [n_kept] the kept raw values in id order, `counts` their frequencies,
    `base` the id of the first of them, `oov` = base + n_kept."""

    def __init__(self, name, keys, counts, base, n_distinct):
        self.name, self.keys, self.counts, self.base = name, keys, counts, int(base)
        self.n_kept, self.n_distinct = int(keys.shape[0]), int(n_distinct)
        self.oov = self.base + self.n_kept

    @property
    def size(self):
        return self.n_kept + 1


def build_field(keys, n_core, base, out_col, name="", capacity=None):
    """Ids of one field.  keys int64 [N] on the device; out_col: an int64 column view (stride = row length) that
    receives the ids.  -> FieldVocab (device tensors)."""
    require_gpu(keys, out_col)
    if keys.dtype != torch.int64 or keys.dim() != 1 or not keys.is_contiguous():
        raise TypeError("build_field: keys must be a contiguous int64 vector")
    N, dev = keys.numel(), keys.device
    if out_col.dtype != torch.int64 or out_col.shape[0] != N or out_col.dim() != 1:
        raise TypeError("build_field: out_col must be an int64 column of N rows")
    if N == 0:
        return FieldVocab(name, torch.empty(0, dtype=torch.int64, device=dev),
                          torch.empty(0, dtype=torch.int32, device=dev), base, 0)
    cap = capacity or max(64, 1 << int(np.ceil(np.log2(2 * N + 1))))
    i32 = dict(dtype=torch.int32, device=dev)
    tkey = torch.empty(cap, dtype=torch.int64, device=dev)
    tcount, tfirst = torch.empty(cap, **i32), torch.empty(cap, **i32)
    slot_of_row, err = torch.empty(N, **i32), torch.zeros(1, **i32)
    check(lib.mapx_vocab_table_init(ptr(tkey), ptr(tcount), ptr(tfirst), cap, stream()))
    check(lib.mapx_vocab_count(ptr(keys), N, ptr(tkey), ptr(tcount), ptr(tfirst), cap, ptr(slot_of_row), ptr(err),
                               stream()))
    slot_entry = torch.empty(cap, **i32)
    ent_cap = min(N, cap)
    ekey = torch.empty(ent_cap, dtype=torch.int64, device=dev)
    ecount, efirst, nm = torch.empty(ent_cap, **i32), torch.empty(ent_cap, **i32), torch.zeros(2, **i32)
    check(lib.mapx_vocab_compact(ptr(tkey), ptr(tcount), ptr(tfirst), cap, ptr(slot_entry), ptr(ekey), ptr(ecount),
                                 ptr(efirst), ptr(nm), stream()))
    e, (U, maxc) = int(err.item()), (int(x) for x in nm.tolist())          # offline path: host syncs are fine
    if e & 1:
        raise ValueError(f"field {name!r}: the value -2^63 is reserved by the vocabulary builder")
    if e & 2:
        raise MapxError(f"field {name!r}: hash table of {cap} slots is full")
    # rank = (count descending, first occurrence ascending): stable sort by first position, then by max - count
    by_first = ops.SegPlan(efirst[:U].contiguous(), N).perm
    keys2 = torch.empty(U, **i32)
    check(lib.mapx_vocab_rank_keys(ptr(ecount), ptr(by_first), U, maxc, ptr(keys2), stream()))
    by_count = ops.SegPlan(keys2, maxc + 1).perm
    rank_of_entry, ranked_counts, n_kept = torch.empty(U, **i32), torch.empty(U, **i32), torch.zeros(1, **i32)
    ranked_keys = torch.empty(U, dtype=torch.int64, device=dev)
    check(lib.mapx_vocab_assign(ptr(by_first), ptr(by_count), ptr(ecount), ptr(ekey), U, int(n_core),
                                ptr(rank_of_entry), ptr(ranked_keys), ptr(ranked_counts), ptr(n_kept), stream()))
    check(lib.mapx_vocab_map(ptr(slot_of_row), ptr(slot_entry), ptr(rank_of_entry), ptr(n_kept), N, int(base),
                             out_col.data_ptr(), out_col.stride(0), stream()))
    k = int(n_kept.item())
    return FieldVocab(name, ranked_keys[:k], ranked_counts[:k], base, U)


def build_vocab(columns, n_core, device="cuda"):
    """columns: ordered {field name: raw column [N]} (the reference's valid_fields without 'click'), host arrays or
    device int64 tensors.  -> (feat_ids int64 [N, F] on the device, [FieldVocab], feat_map {str: id},
    input_size) — `feat_map` as proc_avazu.py:213-250 builds it, `input_size` = len(feat_map)."""
    names = list(columns)
    dec, cols = {}, []
    for name in names:
        c = columns[name]
        if torch.is_tensor(c):
            cols.append(c.to(device=device, dtype=torch.int64).contiguous())
            dec[name] = lambda v: str(int(v))
        else:
            codes, dec[name] = encode_column(c)
            cols.append(torch.as_tensor(codes).to(device))
    N = cols[0].numel() if cols else 0
    if any(c.numel() != N for c in cols):
        raise ValueError("build_vocab: columns of different lengths")
    feat_ids = torch.empty(N, len(names), dtype=torch.int64, device=device)
    feat_map = {tok: i for i, tok in enumerate(RESERVED)}
    fields, base = [], len(RESERVED)
    for f, (name, keys) in enumerate(zip(names, cols)):
        fv = build_field(keys, n_core, base, feat_ids[:, f], name=name)
        for v in fv.keys.cpu().tolist():
            feat_map[f"{name}-{dec[name](v)}"] = len(feat_map)
        feat_map[f"{name}-<oov>"] = len(feat_map)
        assert feat_map[f"{name}-<oov>"] == fv.oov
        fields.append(fv)
        base = fv.oov + 1
    return feat_ids, fields, feat_map, base


def generate_dataset(columns, labels, n_core, data_dir, dataset_name, feat_types=None, seed=42, device="cuda",
                     split=None):
    """The whole of reference generate_dataset() (proc_avazu.py:193-302 / proc_criteo.py:90-196) for raw columns
    held in memory: rows shuffled by numpy's seeded permutation (np.random.seed(42); shuffle(arange(N)) —
    :194-199), vocabulary and ids on the GPU (build_vocab), the reference's meta JSON (index, num_pos, num_neg,
    avg_ctr, field_names, field_map, feat_type_map, feat_map) and the table {feat_ids, field_ids, type_ids,
    labels}, written in the layout code/dataset.py:20-40 reads (here: mapx/dataset.py — `<name>-meta.json`,
    `<name>.npz`).  feat_types: {field: '<cat>' | '<num>'} (default all '<cat>', proc_avazu.py:24); a raw value of
    -1 gets the '<oov>' type (proc_avazu.py:258-261, proc_criteo.py:161-166).
    split: None, or (train, valid, test) fractions for a seeded random `split.pkl` (the reference's own split
    files come from split_criteo_x4.py — scikit-learn's StratifiedKFold — and are used as they are when present)."""
    import json
    import os
    import pickle as pkl
    names = list(columns)
    feat_types = feat_types or {n: "<cat>" for n in names}
    labels = np.asarray(labels)
    np.random.seed(seed)
    index = np.arange(len(labels))
    np.random.shuffle(index)
    labels = labels[index]
    shuffled = {n: np.asarray(columns[n])[index] for n in names}
    field_map, feat_type_map = {"<rsv>": 0}, {"<rsv>": 0}
    for n in names:
        field_map[n] = len(field_map)
        if feat_types[n] not in feat_type_map:
            feat_type_map[feat_types[n]] = len(feat_type_map)
    if "<oov>" not in feat_type_map:
        feat_type_map["<oov>"] = len(feat_type_map)
    feat_ids, _fields, feat_map, input_size = build_vocab(shuffled, n_core, device=device)
    N = len(labels)
    field_ids = np.tile(np.array([field_map[n] for n in names], dtype=np.int32), (N, 1))
    type_ids = np.empty((N, len(names)), dtype=np.int64)
    for j, n in enumerate(names):
        col = shuffled[n]
        missing = (col == -1) if col.dtype.kind in "iu" else np.zeros(N, dtype=bool)
        type_ids[:, j] = np.where(missing, feat_type_map["<oov>"], feat_type_map[feat_types[n]])
    meta = {"index": index.tolist(), "num_pos": int(labels.sum()), "num_neg": int(N - labels.sum()),
            "avg_ctr": float(labels.mean()) if N else 0.0, "field_names": ["<rsv>"] + names, "field_map": field_map,
            "feat_type_map": feat_type_map, "feat_map": feat_map}
    os.makedirs(data_dir, exist_ok=True)
    with open(os.path.join(data_dir, f"{dataset_name}-meta.json"), "w") as f:
        json.dump(meta, f, ensure_ascii=False)
    np.savez(os.path.join(data_dir, f"{dataset_name}.npz"), feat_ids=feat_ids.cpu().numpy(), field_ids=field_ids,
             type_ids=type_ids, labels=labels.astype(np.int64))
    if split is not None and not os.path.exists(os.path.join(data_dir, "split.pkl")):
        rng = np.random.RandomState(seed)
        perm = rng.permutation(N)
        a, b = int(N * split[0]), int(N * (split[0] + split[1]))
        with open(os.path.join(data_dir, "split.pkl"), "wb") as f:
            pkl.dump({"train_index": np.sort(perm[:a]), "valid_index": np.sort(perm[a:b]),
                      "test_index": np.sort(perm[b:])}, f)
    return meta, input_size
