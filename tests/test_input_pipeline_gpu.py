"""SURVEY 8(f1) on the GPU: the input pipeline that replaces the reference's DataLoader
(code/dataset.py:20-60, code/trainer.py:51-58): the split resident in HBM, an epoch = one device
permutation cut into row-gathered batches, shards for data parallelism, feat-count.pt."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dataset(tmp_path, rows=5000, fields=23, vocab=2000):
    from mapx.dataset import BaseDataset, write_synth_dataset

    class Args:
        data_dir = write_synth_dataset(str(tmp_path / "data"), num_rows=rows, num_fields=fields, vocab=vocab)
        dataset_name, pretrain, pt_type, RFD_replace = "avazu", True, "RFD", "Uniform"
    return BaseDataset(Args()), Args


def test_device_split_epoch_is_a_permutation_of_the_host_split(tmp_path):
    """Every epoch deals every row of the split exactly once, in the order of ONE device permutation
    (reproducible from the generator's seed), batch b = rows perm[b*B : (b+1)*B] bit for bit; the
    unshuffled pass is the host array in order with the reference's ragged last batch
    (drop_last=False, trainer.py:51-58)."""
    from mapx.trainer import DeviceSplit
    ds, _ = _dataset(tmp_path)
    train = ds.get_splited_dataset("train")
    sp = DeviceSplit(train, torch.device(DEV))
    assert sp.X.is_cuda and sp.X.dtype == torch.int64 and torch.equal(sp.X.cpu(), torch.from_numpy(train.X))
    B = 512
    gen = torch.Generator(device=DEV).manual_seed(42)
    got_x = torch.cat([x for x, _ in sp.batches(B, True, gen)])
    gen2 = torch.Generator(device=DEV).manual_seed(42)
    perm = torch.randperm(sp.n, device=DEV, generator=gen2)
    assert torch.equal(torch.sort(perm).values, torch.arange(sp.n, device=DEV))
    assert torch.equal(got_x, sp.X[perm])
    got_y = torch.cat([y for _, y in sp.batches(B, True, torch.Generator(device=DEV).manual_seed(42))])
    assert torch.equal(got_y, sp.Y[perm])
    second = torch.cat([x for x, _ in sp.batches(B, True, gen)])             # next epoch: a new permutation
    assert not torch.equal(second, got_x) and torch.equal(torch.sort(second.sum(1)).values, torch.sort(got_x.sum(1)).values)
    plain = [x for x, _ in sp.batches(B, False)]
    assert [len(x) for x in plain] == [B] * (sp.n // B) + ([sp.n % B] if sp.n % B else [])
    assert torch.equal(torch.cat(plain).cpu(), torch.from_numpy(train.X))
    assert sp.num_batches(B) == len(plain)


def test_device_split_shards_are_disjoint_full_batches(tmp_path):
    from mapx.trainer import DeviceSplit
    ds, _ = _dataset(tmp_path)
    sp = DeviceSplit(ds.get_splited_dataset("train"), torch.device(DEV))
    B, W = 256, 4
    shards = []
    for r in range(W):
        gen = torch.Generator(device=DEV).manual_seed(7)               # every rank draws the same permutation
        shards.append([x for x, _ in sp.batches(B, True, gen, (r, W))])
    n = sp.num_batches(B, W)
    assert all(len(s) == n and all(len(x) == B for x in s) for s in shards)
    perm = torch.randperm(sp.n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7))
    for r in range(W):
        for i, x in enumerate(shards[r]):
            b = r + i * W
            assert torch.equal(x, sp.X[perm[b * B:(b + 1) * B]])


def test_feat_count_file_and_field_ranges(tmp_path):
    """feat-count.pt == np.bincount over the train split (dataset.py:49-60); per-field id ranges for
    RFD-Uniform (dataset.py:64-75) contain every id of the field; the device RFD-Uniform generator
    draws inside them."""
    from mapx import ops
    ds, Args = _dataset(tmp_path)
    path = os.path.join(Args.data_dir, "feat-count.pt")
    want = np.bincount(ds.X["train"].reshape(-1), minlength=len(ds.feat_map)).astype(np.float32)
    assert np.array_equal(torch.load(path).numpy(), want) and np.array_equal(ds.feat_count.numpy(), want)
    split = pickle.load(open(os.path.join(Args.data_dir, "split.pkl"), "rb"))
    assert len(split["train_index"]) == len(ds.X["train"])
    lo, hi = ds.idx_low.numpy(), ds.idx_high.numpy()
    allx = np.concatenate([ds.X[s] for s in ("train", "valid", "test")])
    assert bool(((allx >= lo[None, :]) & (allx < hi[None, :])).all())
    ids = torch.from_numpy(ds.X["train"][:1024]).to(DEV)
    L = 6
    out, labels, mi = ops.dynamic_mask_rfd(ids, L, seed=3, offset=9, mode="Uniform", idx_low=ds.idx_low.to(DEV),
                                           idx_high=ds.idx_high.to(DEV), vocab=len(ds.feat_map))
    o = out.cpu().numpy()
    assert bool(((o >= lo[None, :]) & (o < hi[None, :])).all())
    assert torch.equal(labels > 0, out != ids)
