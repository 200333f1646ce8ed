"""Host-side surface, CPU only: flag names/defaults pinned against the reference dataclasses
(golden flag_surface.json), HfArgumentParser-style parsing, Config round trip, state_dict
manifests (golden), LR schedule vs transformers (golden), decay groups, dataset loader."""
import json
import os

import numpy as np
import pytest
import torch

import paramgen as pg
from util import GOLD, make_config


def test_flag_surface_matches_reference():
    from mapx import arguments as A
    gold = json.load(open(os.path.join(GOLD, "flag_surface.json")))
    mine = {}
    for cls, flags in (("ModelArguments", A.MODEL_FLAGS), ("TrainingArguments", A.TRAINING_FLAGS)):
        for name, typ, default, _ in flags:
            mine[name] = dict(cls=cls, required=default is A.REQUIRED,
                              default=None if default is A.REQUIRED else default)
    assert set(mine) == set(gold)
    for k, g in gold.items():
        assert mine[k]["cls"] == g["cls"], k
        assert mine[k]["required"] == g["required"], k
        assert mine[k]["default"] == g["default"], (k, mine[k]["default"], g["default"])


def test_parser_accepts_reference_command_lines():
    from mapx.arguments import parse_args_into_dataclasses
    # run_script/run_DCNv2_MFP.sh
    m, t = parse_args_into_dataclasses(
        "--pretrain=True --output_dir=o --dataset_name=avazu --data_dir=data/avazu --num_train_epochs=3 "
        "--per_gpu_train_batch_size=4096 --per_gpu_eval_batch_size=4096 --learning_rate=1e-3 --lr_sched=cosine "
        "--weight_decay=5e-2 --pt_type=MFP --sampling_method=randint --mask_ratio=0.3 --pt_neg_num=25 "
        "--proj_size=32 --model_name=DCNv2 --embed_size=16 --hidden_size=1000 --num_hidden_layers=3 "
        "--num_cross_layers=3 --hidden_dropout_rate=0.0".split())
    assert t.pretrain is True and t.pt_type == "MFP" and t.learning_rate == 1e-3 and t.mask_ratio == 0.3
    assert m.model_name == "DCNv2" and m.hidden_size == 1000 and m.pt_neg_num == 25
    # run_DCNv2_finetune.sh: bare boolean flag
    m, t = parse_args_into_dataclasses(["--finetune", "--pretrained_model_path", "x/9.model", "--output_dir", "o",
                                        "--model_name", "DCNv2", "--pretrain=False"])
    assert t.finetune is True and t.pretrain is False and t.pretrained_model_path == "x/9.model"
    with pytest.raises(SystemExit):
        parse_args_into_dataclasses(["--output_dir=o"])              # model_name is required
    with pytest.raises(SystemExit):
        parse_args_into_dataclasses(["--model_name=DCNv2", "--output_dir=o", "--no_such_flag=1"])


def test_config_json_round_trip(tmp_path):
    from mapx.arguments import Config
    c = make_config(pg.CASES["A_f23_b7"], "MFP", feat_count=np.ones(1000, dtype=np.float32))
    c.save(str(tmp_path))
    d = Config.load(str(tmp_path))
    assert d.num_fields == 23 and d.pt_type == "MFP" and not hasattr(d, "feat_count")


@pytest.mark.parametrize("mode", ["MFP", "RFD", "CTR"])
@pytest.mark.parametrize("case,backbone", [(c, "DCNv2") for c in pg.CASES] + [("B_f25_b64", b) for b in pg.BACKBONES[1:]])
def test_state_dict_layout_matches_reference(case, backbone, mode):
    from mapx.models import BaseModel
    cfg = pg.CASES[case]
    inp = pg.make_inputs(case, cfg)
    model = BaseModel.from_config(make_config(cfg, mode, inp["feat_count"], backbone=backbone))
    got = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
    key = f"{case}_{mode}" if backbone == "DCNv2" else f"{case}_{mode}_{backbone}"
    gold = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))[key]
    assert got == gold


@pytest.mark.parametrize("mode", ["MFP", "RFD", "CTR"])
def test_state_dict_layout_of_xdeepfm_without_the_mlp_tower_matches_reference(mode):
    from mapx.models import BaseModel
    case, backbone = "B_f25_b64", "xDeepFMCin"
    cfg = pg.CASES[case]
    inp = pg.make_inputs(case, cfg)
    model = BaseModel.from_config(make_config(cfg, mode, inp["feat_count"], backbone=backbone))
    got = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
    gold = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))[f"{case}_{mode}_{backbone}"]
    assert got == gold and not any(k.startswith("dnn.") for k in got)


def test_state_dict_layout_of_autoint_with_lr_and_dnn_matches_reference():
    """The finetune-only modules of AutoInt (lr_layer, dnn, dnn_out; models.py:463-471) under the reference's names."""
    from mapx.models import BaseModel
    case, backbone = "B_f25_b64", "AutoIntFull"
    cfg = pg.CASES[case]
    model = BaseModel.from_config(make_config(cfg, "CTR", None, backbone=backbone))
    got = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
    gold = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))[f"{case}_CTR_{backbone}"]
    assert got == gold
    assert {"lr_layer.embed_w.weight", "lr_layer.bias", "dnn.dnn.0.weight", "dnn.dnn.3.bias", "dnn_out.weight"} <= set(got)


def test_index_linear_buffers_and_init_match_reference():
    from mapx.models import BaseModel
    case = "B_f25_b64"
    cfg = pg.CASES[case]
    inp = pg.make_inputs(case, cfg)
    z = np.load(os.path.join(GOLD, f"{case}_MFP.npz"))
    torch.manual_seed(1)
    model = BaseModel.from_config(make_config(cfg, "MFP", inp["feat_count"]))
    crit = model.mfp_criterion
    np.testing.assert_allclose(crit.logprob_noise.numpy(), z["nce/logprob_noise"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(crit.alias.prob.numpy(), z["nce/alias_prob"])
    assert np.array_equal(crit.alias.alias.numpy(), z["nce/alias_alias"])
    np.testing.assert_allclose(crit.bias.weight.detach().numpy(), z["nce/bias_init"], rtol=1e-6, atol=1e-6)
    assert float(crit.emb.weight.abs().max()) <= 1 / np.sqrt(cfg["P"]) + 1e-7
    std = float(model.embed.embedding.weight.std())
    assert std == pytest.approx(np.sqrt(2.0 / (cfg["F"] + cfg["E"])), rel=0.05)


def test_alias_cache_files_round_trip(tmp_path):
    """data_dir/alias_self_{prob,alias}.h5 are torch.save files, as upstream."""
    from mapx.models import BaseModel
    cfg = pg.CASES["A_f23_b7"]
    inp = pg.make_inputs("A_f23_b7", cfg)
    c = make_config(cfg, "MFP", inp["feat_count"], data_dir=str(tmp_path))
    m1 = BaseModel.from_config(c)
    assert os.path.exists(tmp_path / "alias_self_prob.h5") and os.path.exists(tmp_path / "alias_self_alias.h5")
    assert torch.load(tmp_path / "alias_self_alias.h5").dtype == torch.int64
    m2 = BaseModel.from_config(c)                      # second construction loads the cache
    assert torch.equal(m1.mfp_criterion.alias.alias, m2.mfp_criterion.alias.alias)


def test_lr_lambda_matches_transformers_golden():
    from mapx.optim import lr_lambda
    z = np.load(os.path.join(GOLD, "lr_schedules.npz"))
    for name in z.files:
        kind, T, W = name.split("_")
        kind = "cosine" if kind == "cos" else "const"
        got = [lr_lambda(kind, s, int(T[1:]), int(W[1:])) for s in range(len(z[name]))]
        np.testing.assert_allclose(got, z[name], rtol=1e-12, atol=1e-15)
    with pytest.raises(NotImplementedError):
        lr_lambda("linear", 0, 10, 0)


def test_decay_rule_is_the_reference_name_rule():
    from mapx.optim import decays
    assert decays("embed.embedding.weight") and decays("mfp_criterion.emb.weight")
    assert not decays("mfp_criterion.bias.weight") and not decays("feat_encoder.bias")


def test_dataset_loader_and_feat_count(tmp_path):
    from mapx.arguments import TrainingArguments
    from mapx.dataset import BaseDataset, write_synth_dataset
    d = write_synth_dataset(str(tmp_path / "avazu"), num_rows=500, num_fields=23, vocab=600)
    args = TrainingArguments(output_dir="o", data_dir=d, dataset_name="avazu", pretrain=True)
    ds = BaseDataset(args)
    assert ds.X["train"].shape == (400, 23) and ds.X["valid"].shape == (50, 23) and ds.X["test"].shape == (50, 23)
    assert len(ds.feat_map) == 600 and len(ds.field_map) - 1 == 23
    cnt = np.bincount(ds.X["train"].reshape(-1), minlength=600)
    assert torch.equal(ds.feat_count, torch.from_numpy(cnt.astype(np.float32)))
    assert os.path.exists(os.path.join(d, "feat-count.pt"))
    ds2 = BaseDataset(args)                               # second run loads the cached file
    assert torch.equal(ds.feat_count, ds2.feat_count)
    with pytest.raises(AssertionError):
        ds.get_splited_dataset("dev")
    x, y = ds.get_splited_dataset("train")[3]
    assert x.shape == (23,)


def test_hdf5_tables_read_without_h5py(tmp_path):
    """The reference's <name>.h5 (dataset.py:27-29) loads through libhdf5 + ctypes when h5py is
    absent.  Fixture: a real HDF5 file written by the HDF5 C library (tests/golden/gen_h5_fixture.py):
    contiguous int64 tables as proc_avazu.py writes them, one chunked+gzip and one big-endian set."""
    import json
    import pickle
    import shutil
    from mapx import h5lite
    try:
        h5lite.library()
    except ImportError as e:
        pytest.skip(str(e))
    want = np.load(os.path.join(GOLD, "tiny_table_expected.npz"))
    got = h5lite.read_datasets(os.path.join(GOLD, "tiny_table.h5"), list(want.files))
    for k in want.files:
        assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
    with pytest.raises(KeyError):
        h5lite.read_datasets(os.path.join(GOLD, "tiny_table.h5"), ["absent"])
    with pytest.raises(OSError):
        h5lite.read_datasets(os.path.join(GOLD, "tiny_table_expected.npz"), ["feat_ids"])
    # the loader itself, on a data_dir that only holds the .h5
    from mapx.arguments import TrainingArguments
    from mapx.dataset import BaseDataset
    d = tmp_path / "tiny"
    d.mkdir()
    shutil.copy(os.path.join(GOLD, "tiny_table.h5"), d / "tiny.h5")
    N, F = want["feat_ids"].shape
    meta = dict(field_names=[f"C{i}" for i in range(F)], feat_map={str(i): i for i in range(1000)},
                field_map={f"C{i}": i for i in range(F)})
    (d / "tiny-meta.json").write_text(json.dumps(meta))
    idx = np.arange(N)
    with open(d / "split.pkl", "wb") as f:
        pickle.dump(dict(train_index=idx[:48], valid_index=idx[48:56], test_index=idx[56:]), f)
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py present: the loader prefers it")
    except ImportError:
        pass
    ds = BaseDataset(TrainingArguments(output_dir="o", data_dir=str(d), dataset_name="tiny", pretrain=True))
    assert np.array_equal(ds.X["train"], want["feat_ids"][:48]) and np.array_equal(ds.Y["test"], want["labels"][56:])
    assert float(ds.feat_count.sum()) == 48 * F


def test_no_cpu_path():
    from mapx.arguments import TrainingArguments
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        TrainingArguments(output_dir="o").device


def test_optimizer_state_slot_size_is_inferred_when_the_key_is_missing():
    """dense moments are saved as flat buffers whose per-parameter slots were 4, then 8 elements before the
    state recorded `flat_pad`.  A keyless state in slots of 8 that holds a one-element parameter (the bias of a
    one-output layer) does not fit slots of 4: the slot size is the one that fits, 8 tried first; nothing fits: ValueError."""
    import pytest
    import torch
    try:
        from mapx.optim import MapxOptimizer
    except ImportError as e:          # the HIP library is not built here
        pytest.skip(str(e))
    numels = [736, 1, 23, 64]

    def flat(pad, fill):
        parts = []
        for i, n in enumerate(numels):
            piece = torch.zeros((n + pad - 1) // pad * pad)
            piece[:n] = fill + i + torch.arange(n) * 1e-3
            parts.append(piece)
        return torch.cat(parts)

    def blank():
        opt = object.__new__(MapxOptimizer)
        opt.done, opt.steps_done, opt.tables, opt.bf16 = torch.zeros(1, dtype=torch.int32), 0, [], False
        opt.groups = [dict(names=["a", "b", "c", "d"], numels=numels, m=torch.zeros_like(flat(8, 0.0)),
                           v=torch.zeros_like(flat(8, 0.0)))]
        return opt

    for pad in (8, 4):
        opt = blank()
        sd = dict(steps_done=3, done=torch.tensor([3], dtype=torch.int32), tables=[],
                  groups=[dict(names=["a", "b", "c", "d"], m=flat(pad, 1.0), v=flat(pad, 2.0))])
        opt.load_state_dict(sd)
        assert torch.equal(opt.groups[0]["m"], flat(8, 1.0)) and torch.equal(opt.groups[0]["v"], flat(8, 2.0)), pad
    opt = blank()
    sd["groups"][0]["m"] = sd["groups"][0]["m"][:-1]
    sd["groups"][0]["v"] = sd["groups"][0]["v"][:-1]
    with pytest.raises(ValueError):
        opt.load_state_dict(sd)


def test_vocab_hex_path_needs_every_string_to_be_fixed_width_lower_case_hex():
    """encode_column's hexadecimal fast path (exact integer codes for the 8-digit hashes of Avazu / Criteo)
    looked at the first 64 strings only; '1A' vs '1a', ' 1f' vs '01f' or '1_0' all parse with int(s, 16) and would have
    merged into one vocabulary entry where the reference's Counter (proc_avazu.py:237-251, keyed by the raw string)
    keeps them apart.  Every string must match [0-9a-f]{w}; anything else is numbered by first occurrence."""
    import numpy as np
    import pytest
    try:
        from mapx.vocab import encode_column
    except ImportError as e:
        pytest.skip(str(e))
    codes, dec = encode_column(np.array(["0a1f", "ffff", "0a1f"] + ["00ff"] * 70))
    assert codes[0] == codes[2] == 0x0a1f and codes[1] == 0xffff and dec(codes[0]) == "0a1f"
    late = ["0a1f"] * 70 + ["0A1F", " a1f", "1_0f", "+a1f"]            # the odd ones come after the first 64
    codes, dec = encode_column(np.array(late))
    assert len(set(codes.tolist())) == 5 and [dec(c) for c in codes[-4:]] == late[-4:]
