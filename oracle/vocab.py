"""ORACLE — test infrastructure only, never the product path.

Plain-Python restatement of the per-field vocabulary / id assignment of the reference's offline
preprocessing, the arithmetic mapx/vocab.py + csrc/vocab.hip reproduce on the GPU:

  reserved ids      data_preprocess/proc_avazu.py:213-220  ( = proc_criteo.py:109-116 )
  per-field ranking proc_avazu.py:247-250                  ( = proc_criteo.py:149-152 )
                    Counter(feat).most_common(): descending count, equal counts in order of first
                    occurrence (Counter keeps insertion order, most_common() sorts stably); values with
                    count >= n_core get the next ids, then one <oov> id per field
  row translation   proc_avazu.py:252-257                  ( = proc_criteo.py:155-160 )

Parity pin: tests/golden/vocab_avazu.npz, vocab_criteo.npz — the feat_map that the reference's own
generate_dataset() wrote for seeded synthetic columns (tests/golden/gen_vocab_golden.py), checked by
tests/test_oracle_golden.py.  Only tests/ may import this module.
"""
from collections import Counter

RESERVED = ("<pad>", "<cls>", "<sep>", "<mask>") + tuple(f"<unused{i}>" for i in range(6))


def build_feat_map(columns, n_core):
    """columns: ordered {field name: sequence of raw values}.  -> (feat_map {str: id}, feat_ids rows-major list
    of lists [N][F])."""
    feat_map = {tok: i for i, tok in enumerate(RESERVED)}
    per_field = []
    for name, feat in columns.items():
        feat = list(feat)
        for k, v in Counter(feat).most_common():
            if v >= n_core:
                feat_map[f"{name}-{k}"] = len(feat_map)
        oov = feat_map[f"{name}-<oov>"] = len(feat_map)
        per_field.append([feat_map.get(f"{name}-{f}", oov) for f in feat])
    n = len(per_field[0]) if per_field else 0
    return feat_map, [[col[i] for col in per_field] for i in range(n)]
