"""Cross-checks the fp32 BERT restatement against transformers.BertModel (third-party
code present in the image, not the reference) and pins a toy-config golden.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import bert as OB


def _hf_model(cfg, w):
    from transformers import BertConfig, BertModel
    hf = BertModel(BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden,
                              num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                              intermediate_size=cfg.inter, max_position_embeddings=cfg.max_pos,
                              type_vocab_size=cfg.type_vocab, hidden_act="gelu",
                              layer_norm_eps=cfg.ln_eps, hidden_dropout_prob=0.0,
                              attention_probs_dropout_prob=0.0), add_pooling_layer=False)
    missing, unexpected = hf.load_state_dict(w, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing)
    return hf.eval()


def test_weight_inventory_matches_hf_bert_large():
    cfg = OB.BertCfg()
    names = OB.weight_names(cfg)
    assert len(names) == 5 + 24 * 16 == 389
    n_params = sum(int(np.prod(OB.weight_shape(cfg, n))) for n in names)
    assert n_params == 334_092_288


def test_toy_forward_matches_transformers(golden_dir):
    cfg = OB.BertCfg.toy()
    w = OB.random_weights(cfg, seed=3)
    rng = np.random.default_rng(0)
    ids = rng.integers(5, cfg.vocab_size, (4, 24))
    lens = np.array([24, 7, 1, 16])
    cls, hidden = OB.bert_encode(w, cfg, ids, lens, return_hidden=True)
    hf = _hf_model(cfg, w)
    mask = (np.arange(24)[None, :] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        out = hf(input_ids=torch.as_tensor(ids), attention_mask=torch.as_tensor(mask)).last_hidden_state
    ref = out[:, 0].numpy()
    assert np.allclose(cls, ref, atol=2e-5, rtol=1e-5)
    for b in range(4):                                    # valid token rows agree everywhere
        assert np.allclose(hidden[-1][b, :lens[b]], out[b, :lens[b]].numpy(), atol=2e-5, rtol=1e-5)
    path = os.path.join(golden_dir, "bert_toy.npz")
    if os.path.exists(path):
        g = np.load(path)
        assert np.array_equal(g["ids"], ids) and np.array_equal(g["lens"], lens)
        assert np.allclose(g["cls"], cls, atol=1e-5)


def test_padding_is_ignored():
    cfg = OB.BertCfg.toy()
    w = OB.random_weights(cfg, seed=1)
    rng = np.random.default_rng(1)
    ids = rng.integers(5, cfg.vocab_size, (2, 16))
    lens = np.array([9, 16])
    a = OB.bert_encode(w, cfg, ids, lens)
    ids2 = ids.copy(); ids2[0, 9:] = 0
    b = OB.bert_encode(w, cfg, ids2, lens)
    c = OB.bert_encode(w, cfg, ids[:1, :9], lens[:1])
    assert np.allclose(a, b, atol=1e-6) and np.allclose(a[0], c[0], atol=2e-6)
