"""SURVEY §8 f4 on the GPU: the vocabulary builders (mapx/vocab.py over csrc/vocab.hip) against the feat_map
and feat_ids of the reference's own preprocessing (tests/golden/vocab_*.npz, from
data_preprocess/proc_avazu.py / proc_criteo.py generate_dataset()), against the oracle on larger skewed
columns, and through size-independent properties at the size of a real Avazu column.  Integer work: bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("name", ["avazu", "criteo"])
def test_vocab_matches_reference_fixture(golden_dir, name):
    from mapx import vocab
    z = np.load(os.path.join(golden_dir, f"vocab_{name}.npz"))
    cols = {str(n): z[f"col/{n}"] for n in z["names"]}
    feat_ids, fields, feat_map, input_size = vocab.build_vocab(cols, int(z["n_core"]), device=DEV)
    assert input_size == int(z["input_size"]) == len(feat_map)
    assert list(feat_map.keys()) == [str(k) for k in z["feat_map_keys"]]           # same names in the same order
    assert list(feat_map.values()) == z["feat_map_ids"].tolist()
    assert np.array_equal(feat_ids.cpu().numpy(), z["feat_ids"])
    for fv, n in zip(fields, z["names"]):
        assert feat_map[f"{n}-<oov>"] == fv.oov and fv.size == fv.n_kept + 1


@pytest.mark.parametrize("n_rows,vocab_size,n_core", [(1, 5, 1), (63, 3, 2), (5000, 40000, 1), (200_000, 50_000, 5),
                                                       (300_001, 7, 1000), (10_000, 10_000, 3)])
def test_vocab_field_vs_oracle(n_rows, vocab_size, n_core):
    """One field against the Counter restatement: ties in the counts resolved by first occurrence, values below
    n_core folded into <oov>, negative and 2^40-sized raw values, a column where nothing / everything is kept."""
    from mapx import vocab
    from oracle import vocab as V
    rng = np.random.RandomState(n_rows % 9973 + vocab_size)
    raw = rng.randint(-(1 << 40), 1 << 40, size=vocab_size).astype(np.int64)
    raw[: min(3, vocab_size)] = [-1, 0, 2 ** 62][: min(3, vocab_size)]
    ranks = np.minimum((rng.pareto(0.8, n_rows) * 3).astype(np.int64), vocab_size - 1)
    col = raw[ranks]
    out = torch.full((n_rows, 3), -7, dtype=torch.int64, device=DEV)
    fv = vocab.build_field(torch.from_numpy(col).to(DEV), n_core, 10, out[:, 1], name="f")
    feat_map, rows = V.build_feat_map({"f": col.tolist()}, n_core)
    want = np.array(rows, dtype=np.int64)[:, 0]
    assert np.array_equal(out[:, 1].cpu().numpy(), want)
    assert bool((out[:, 0] == -7).all()) and bool((out[:, 2] == -7).all())          # a column view: neighbours untouched
    kept = [k for k in feat_map if k.startswith("f-") and k != "f-<oov>"]
    assert [f"f-{v}" for v in fv.keys.cpu().tolist()] == kept and fv.oov == feat_map["f-<oov>"]
    assert fv.n_distinct == len(set(col.tolist()))
    cnt = fv.counts.cpu().numpy()
    assert bool((cnt >= n_core).all()) and bool((np.diff(cnt) <= 0).all())


def test_vocab_reserved_value_and_empty_column():
    from mapx import vocab
    out = torch.empty(4, 1, dtype=torch.int64, device=DEV)
    with pytest.raises(ValueError):
        vocab.build_field(torch.tensor([1, -2 ** 63, 3, 1], device=DEV), 1, 10, out[:, 0])
    fv = vocab.build_field(torch.empty(0, dtype=torch.int64, device=DEV), 1, 10, torch.empty(0, dtype=torch.int64, device=DEV))
    assert fv.n_kept == 0 and fv.oov == 10


def test_vocab_hex_strings_and_decode():
    """Columns of 8-digit hexadecimal hashes (the raw Avazu / Criteo ids) parse to integers exactly and print back
    as the reference's f-string would; other strings are coded by first occurrence."""
    from mapx import vocab
    from oracle import vocab as V
    col_hex = np.array(["a99f214a", "0000000f", "a99f214a", "ffffffff", "0000000f", "a99f214a"])
    col_txt = np.array(["Mon", "tue", "Mon", "Mon", "x-y", "tue"])
    ids, fields, feat_map, size = vocab.build_vocab({"site_id": col_hex, "day": col_txt}, 2, device=DEV)
    ref_map, rows = V.build_feat_map({"site_id": col_hex.tolist(), "day": col_txt.tolist()}, 2)
    assert feat_map == ref_map and list(feat_map) == list(ref_map) and size == len(ref_map)
    assert ids.cpu().tolist() == rows


def test_vocab_full_size_column_properties():
    """At the size of one real Avazu column (40 M rows, millions of distinct values; the oracle's Python Counter
    would need minutes): ids stay inside [base, oov]; kept values' ids are ranks by count — the histogram of the
    produced ids is non-increasing over the kept range and equals the reported counts; every row of a kept value
    maps to that value's one id; rows of dropped values are exactly the <oov> rows; idempotence: building again
    from the produced ids reproduces them up to the base offset."""
    from mapx import vocab
    N, n_core, base = 40_000_000, 5, 10
    g = torch.Generator(device=DEV).manual_seed(3)
    u = torch.rand(N, device=DEV, generator=g)
    raw = (u.pow(6) * 6_700_000).long() * 2654435761 % (1 << 40) - (1 << 39)          # skewed, scrambled values
    out = torch.empty(N, 1, dtype=torch.int64, device=DEV)
    fv = vocab.build_field(raw, n_core, base, out[:, 0], name="device_ip")
    ids = out[:, 0]
    assert int(ids.min()) >= base and int(ids.max()) <= fv.oov
    hist = torch.bincount(ids - base, minlength=fv.n_kept + 1)
    assert torch.equal(hist[:fv.n_kept].to(torch.int32), fv.counts)
    assert bool((hist[:fv.n_kept][1:] <= hist[:fv.n_kept][:-1]).all()) and int(hist[:fv.n_kept].min()) >= n_core
    # value -> id is a function, id -> value its inverse on the kept range
    kept_rows = ids < fv.oov
    assert torch.equal(fv.keys[(ids[kept_rows] - base)], raw[kept_rows])
    uniq, cnt = torch.unique(raw[~kept_rows], return_counts=True)
    assert int(cnt.max()) < n_core and fv.n_distinct == fv.n_kept + uniq.numel()
    out2 = torch.empty(N, 1, dtype=torch.int64, device=DEV)
    fv2 = vocab.build_field(ids, n_core, base, out2[:, 0])
    # (ids are ranks already; <oov> is one more value, ranked by its own count among them)
    assert fv2.n_kept in (fv.n_kept, fv.n_kept + 1)
    assert torch.equal(torch.bincount(out2[:, 0] - base).sort(descending=True).values[:fv.n_kept],
                       torch.bincount(ids - base).sort(descending=True).values[:fv.n_kept])


def test_generate_dataset_is_read_back_by_the_loader(golden_dir, tmp_path):
    """Raw columns -> (shuffle, vocabulary, ids, meta JSON, table) -> mapx.dataset.BaseDataset: the reference's row
    permutation (np.random.seed(42) + shuffle) and feat_map, the loader's splits and feat-count file."""
    from mapx import vocab
    from mapx.dataset import BaseDataset
    z = np.load(os.path.join(golden_dir, "vocab_avazu.npz"))
    names = [str(n) for n in z["names"]]
    N = z["feat_ids"].shape[0]
    # the fixture's columns are in the reference's SHUFFLED order: undo its permutation to get "raw" columns
    np.random.seed(42)
    index = np.arange(N)
    np.random.shuffle(index)
    raw = {n: np.empty_like(z[f"col/{n}"]) for n in names}
    for n in names:
        raw[n][index] = z[f"col/{n}"]
    labels = (np.arange(N) % 6 == 0).astype(np.int64)
    meta, input_size = vocab.generate_dataset(raw, labels, int(z["n_core"]), str(tmp_path), "avazu", split=(0.8, 0.1, 0.1))
    assert meta["index"] == index.tolist() and input_size == int(z["input_size"])
    assert list(meta["feat_map"].keys()) == [str(k) for k in z["feat_map_keys"]]
    assert meta["field_names"] == ["<rsv>"] + names and meta["field_map"]["weekday"] == 1
    tab = np.load(os.path.join(str(tmp_path), "avazu.npz"))
    assert np.array_equal(tab["feat_ids"], z["feat_ids"]) and np.array_equal(tab["labels"], labels[index])

    class Args:
        data_dir, dataset_name, pretrain, pt_type, RFD_replace = str(tmp_path), "avazu", True, "MFP", "Unigram"
    ds = BaseDataset(Args())
    tr = ds.get_splited_dataset("train")
    assert tr.X.shape == (int(N * 0.8), len(names)) and ds.feat_count.shape[0] >= input_size - 1
