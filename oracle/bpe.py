"""Oracle (test infrastructure): byte-level BPE of openai/CLIP's ``clip/simple_tokenizer.py`` (third-party, pinned by
requirements.txt:17 to openai/CLIP @ d50d76daa670286dd6cacf3bcd80b5e4823fc8e1, absent from /root/reference), restated the way the
published file writes it: a ``get_pairs`` set, ``min`` over ranks, ``word.index`` scanning.  Parity status: UNPINNED by reference-held
vectors (the vocabulary file cannot be fetched here); the algorithm is checked on a synthetic merges table and hand-derived ids in
tests/test_tokenizer.py, and the call-site contract ([SOT] + ids + [EOT], zero padding to 77) against label_reward.py:136-138 /
arp_dt/models/openai/tokenizer.py:19-41."""
import html

import regex as re


def bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, map(chr, cs)))


def get_pairs(word):
    return {(a, b) for a, b in zip(word, word[1:])}


class Tokenizer:
    def __init__(self, merges):
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.bpe_ranks = dict(zip([tuple(m) for m in merges], range(len(merges))))
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}  # the published file pre-seeds its cache so
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", re.IGNORECASE)

    def bpe(self, token):
        if token in self.cache:  # ... so the literal special strings map to the special ids
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = get_pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda pair: self.bpe_ranks.get(pair, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new_word = []
            i = 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                    new_word.extend(word[i:j])
                    i = j
                except ValueError:
                    new_word.extend(word[i:])
                    break
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new_word.append(first + second)
                    i += 2
                else:
                    new_word.append(word[i])
                    i += 1
            word = tuple(new_word)
            if len(word) == 1:
                break
            pairs = get_pairs(word)
        return " ".join(word)

    def encode(self, text):
        text = re.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip().lower()
        out = []
        for token in re.findall(self.pat, text):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            out.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return out
