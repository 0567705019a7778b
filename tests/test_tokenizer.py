"""SURVEY row L3 (clip.tokenize): the product's byte-level BPE against the oracle restatement of openai/CLIP's simple_tokenizer
on a synthetic merges table (the real vocabulary file cannot be fetched here), hand-derived ids, and the call-site contract."""
import gzip
import os

import numpy as np
import pytest

from arp_amd import data
from arp_amd import tokenizer as T


def _merges():
    """A small rank-ordered merges table built the way BPE training would: frequent pairs of the task prompts first."""
    words = ("the goal is to collect the coin . navigate a maze to collect the yellow cheese line red gem "
             "agent must go far right of level his voice echoed through empty hallway").split()
    merges, seen = [], set()
    for w in words:
        sym = list(w[:-1]) + [w[-1] + "</w>"]
        while len(sym) > 1:  # left-to-right chain: every prefix of every word becomes a merge
            m = (sym[0], sym[1])
            if m not in seen:
                seen.add(m)
                merges.append(m)
            sym = [sym[0] + sym[1]] + sym[2:]
    return merges


def test_ids_by_hand():
    tok = T.SimpleTokenizer(merges=[("t", "h"), ("th", "e</w>"), ("c", "o"), ("co", "i"), ("coi", "n</w>")])
    b2u = T.bytes_to_unicode()
    assert len(tok.encoder) == 512 + 5 + 2 and tok.encoder["<|startoftext|>"] == 517 and tok.encoder["<|endoftext|>"] == 518
    # "the" -> one merged token (id 512 + 1); "coin" -> id 512 + 4; "." -> the </w> variant of the byte symbol
    dot = 256 + list(b2u.values()).index(".")
    assert tok.encode("The  coin.") == [513, 516, dot]
    # an unknown word falls apart into byte symbols, the last one in its </w> form
    xs = tok.encode("zq")
    assert xs == [list(b2u.values()).index("z"), 256 + list(b2u.values()).index("q")]
    # UTF-8 bytes outside ASCII go through the printable alphabet
    assert len(tok.encode("é")) == 2 and tok.decode(tok.encode("the coin")).strip() == "the coin"


def test_matches_oracle_on_prompts_and_noise():
    from oracle import bpe as O
    merges = _merges()
    tok, ora = T.SimpleTokenizer(merges=merges), O.Tokenizer(merges)
    texts = [data.get_clip_instruct(t) for t in ("coinrun", "maze", "maze_yellowline", "maze_redline_yellowgem")]
    texts += [data.get_clip_special_instruct("coinrun", k) for k in ("random1", "random2", "misinfo", "misinfo2", "misinfo3", "misinfo4")]
    texts += ["it's the agent's goal, isn't it?  I'll go -- 42 coins & more;  &amp;lt;tag&amp;gt;", "   ", "<|startoftext|>the<|endoftext|>", "naïve café №5"]
    rng = np.random.default_rng(0)
    alphabet = list("abcdefghijklmnopqrstuvwxyz      '.,!?0123456789-THEGOAL")
    texts += ["".join(rng.choice(alphabet, int(rng.integers(1, 60)))) for _ in range(200)]
    for t in texts:
        assert tok.encode(t) == ora.encode(t), t


def test_tokenize_contract_and_file_loading(tmp_path):
    merges = _merges()
    # the published file: a header line, then one merge per line (gzip); extra lines beyond the kept count are ignored
    p = tmp_path / "bpe.txt.gz"
    with gzip.open(p, "wb") as f:
        f.write(("#version: synthetic\n" + "\n".join(" ".join(m) for m in merges) + "\n").encode())
    fn = T.build_tokenizer(str(p))
    out = fn(["the goal is to collect the coin.", "navigate a maze"])
    tok = T.SimpleTokenizer(merges=merges)
    sot, eot = tok.encoder["<|startoftext|>"], tok.encoder["<|endoftext|>"]
    assert out.shape == (2, 77) and out.dtype == np.int32 and out[0, 0] == sot
    n0 = len(tok.encode("the goal is to collect the coin."))
    assert out[0, n0 + 1] == eot and (out[0, n0 + 2 :] == 0).all() and out[0].argmax() == n0 + 1  # EOT is the largest id: the EOT-pool rule
    assert (fn("navigate a maze") == out[1:2]).all()  # a bare string is one prompt
    long = " ".join(["coin"] * 100)
    with pytest.raises(RuntimeError, match="too long"):
        fn(long)
    t = T.tokenize(long, tok, truncate=True)
    assert t.shape == (1, 77) and t[0, -1] == eot and t[0, 0] == sot
    with pytest.raises(ValueError, match="no BPE vocabulary"):
        T.SimpleTokenizer()
    # with the real file's merge count the special ids are CLIP's 49406 / 49407
    assert 512 + T.N_MERGES_KEPT == 49406


def test_plugs_into_label_reward():
    """label_reward(tokenizer=...) is the L3 seam: prompt text in, token ids to the labeller."""
    from arp_amd import label_reward as L
    tok = T.SimpleTokenizer(merges=_merges())
    seen = {}

    class Fake:
        def set_text(self, t):
            seen["tokens"] = np.asarray(t)
            return self

        def label(self, frames, use_crop=False):
            return np.zeros(len(frames), np.float32)

        def close(self):
            pass

    st = {"ob": np.zeros((3, 2, 4, 4, 3), np.uint8), "done": np.array([[0, 0], [0, 0], [0, 1]], np.float32)}
    L.label_reward("coinrun", "hard", 500, 0, data.get_clip_instruct("coinrun"), ".", store=st, clip_model=Fake(),
                   tokenizer=lambda texts: T.tokenize(texts, tok))
    assert seen["tokens"].shape == (1, 77) and seen["tokens"][0, 0] == tok.encoder["<|startoftext|>"]
