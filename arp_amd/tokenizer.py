"""``clip.tokenize`` (label_reward.py:136-138; arp_dt/models/openai/tokenizer.py:19-64 wraps the same class).

The algorithm lives in the third-party package **openai/CLIP @ d50d76daa670286dd6cacf3bcd80b5e4823fc8e1**
(``clip/simple_tokenizer.py``, requirements.txt:17), absent from /root/reference; this restates its published byte-level BPE:

  * text -> ``html.unescape`` twice, strip, collapse whitespace, lower-case (``ftfy.fix_text`` first when the package exists: it
    is a no-op on the ASCII prompts of data_procgen.py:281-317);
  * split with the CLIP pattern (special tokens, English contractions, letter runs, single digits, other non-space runs);
  * every piece: UTF-8 bytes -> the printable ``bytes_to_unicode`` alphabet, last symbol + ``</w>``, then repeatedly merge the
    adjacent pair with the lowest rank in the merges table until none is left;
  * ids: 256 byte symbols, 256 ``</w>`` variants, one per merge, then ``<|startoftext|>``, ``<|endoftext|>``.

With the real ``bpe_simple_vocab_16e6.txt.gz`` (48 894 merges kept) that gives SOT = 49406, EOT = 49407.  The file is NOT in this
image (no network): pass its path (``bpe_path``) where it exists; tests use a small synthetic merges table.
``tokenize`` has ``clip.tokenize``'s contract: int [n, 77], zero padded, RuntimeError when a prompt is too long.
"""
import gzip
import html
from functools import lru_cache

import numpy as np
import regex as re

N_MERGES_KEPT = 49152 - 256 - 2  # simple_tokenizer.py: merges[1 : 49152-256-2+1]


@lru_cache()
def bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _clean(text):
    try:
        import ftfy  # optional: absent in this image
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class SimpleTokenizer:
    def __init__(self, bpe_path=None, merges=None):
        """bpe_path: the merges file (plain or .gz; first line is a header, as in the published file), or ``merges``: a list of
        (left, right) pairs in rank order."""
        if merges is None:
            if bpe_path is None:
                raise ValueError("no BPE vocabulary in this image: pass bpe_path=<bpe_simple_vocab_16e6.txt.gz> or merges=[...]")
            opener = gzip.open if str(bpe_path).endswith(".gz") else open
            with opener(bpe_path, "rb") as f:
                lines = f.read().decode("utf-8").split("\n")
            merges = [tuple(m.split()) for m in lines[1 : N_MERGES_KEPT + 1] if m.strip()]
        merges = [tuple(m) for m in merges]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab += ["".join(m) for m in merges]
        vocab += ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {v: i for i, v in enumerate(vocab)}
        self.decoder = {i: v for v, i in self.encoder.items()}
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", re.IGNORECASE)

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            ranks = [self.bpe_ranks.get((a, b), None) for a, b in zip(word, word[1:])]
            best = min((r for r in ranks if r is not None), default=None)
            if best is None:
                break
            first, second = word[ranks.index(best)], word[ranks.index(best) + 1]
            out, i = [], 0
            while i < len(word):  # every non-overlapping occurrence of the pair, left to right
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    out.append(first + second)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = out
        res = " ".join(word)
        self.cache[token] = res
        return res

    def encode(self, text):
        ids = []
        for token in re.findall(self.pat, _clean(text).lower()):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return ids

    def decode(self, ids):
        text = "".join(self.decoder[int(i)] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


def tokenize(texts, tokenizer, context_length=77, truncate=False):
    """``clip.tokenize`` (clip/clip.py) / ``_tokenize`` (arp_dt/models/openai/tokenizer.py:19-41): [SOT] + ids + [EOT], zero padded."""
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = tokenizer.encoder["<|startoftext|>"], tokenizer.encoder["<|endoftext|>"]
    out = np.zeros((len(texts), context_length), np.int32)
    for i, t in enumerate(texts):
        ids = [sot] + tokenizer.encode(t) + [eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[: context_length - 1] + [eot]
        out[i, : len(ids)] = ids
    return out


def build_tokenizer(bpe_path, truncate=False):
    """arp_dt/models/openai/tokenizer.py:44-64 without the download: texts -> int32 [n, 77]; plugs into
    ``label_reward(..., tokenizer=build_tokenizer(path))``."""
    tok = SimpleTokenizer(bpe_path)
    return lambda texts: tokenize(texts, tok, 77, truncate)
