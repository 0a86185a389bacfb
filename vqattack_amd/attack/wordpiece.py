"""BERT WordPiece tokenisation from a ``vocab.txt`` -- what the file-fed entry points need of a tokenizer.

The reference tokenises with ``BertTokenizer.from_pretrained('bert-base-uncased')`` (``adv_attack.py:51-52``;
``vlmo_module.py`` through its datamodule), i.e. the published BERT algorithm: lower-case + accent stripping,
whitespace split, every punctuation character its own token, then greedy longest-match-first word pieces with a ``##``
continuation prefix and ``[UNK]`` for words with no segmentation (or longer than 100 characters).  Offline there is no
hub to fetch the tokenizer from, so it is restated here over a user-supplied vocabulary file (one token per line, line
number = id); ``tests/test_file_inputs.py`` compares it with the ``tokenizers`` library's ``BertWordPieceTokenizer``.

Only what the attack reads is provided: the attack works on WORDS (whitespace-separated pieces of the question,
``Adv_attack._tokenize`` adv_attack.py:141-154) and their piece ids.
"""
import unicodedata


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPiece:
    def __init__(self, vocab_file, lower=True, unk="[UNK]", max_chars=100):
        with open(vocab_file, encoding="utf-8") as fh:
            tokens = [line.rstrip("\n") for line in fh]
        while tokens and tokens[-1] == "":
            tokens.pop()
        self.vocab = {t: i for i, t in enumerate(tokens)}
        self.tokens = tokens
        self.lower, self.unk, self.max_chars = lower, unk, max_chars
        for name in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"):
            if name not in self.vocab:
                raise ValueError("{}: no {} entry".format(vocab_file, name))
        self.pad_id, self.unk_id, self.cls_id = self.vocab["[PAD]"], self.vocab["[UNK]"], self.vocab["[CLS]"]
        self.sep_id, self.mask_id = self.vocab["[SEP]"], self.vocab["[MASK]"]

    # ---- basic tokenisation ---------------------------------------------------------------------------------------
    def _clean(self, text):
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch) in ("Cc", "Cf") and ch not in "\t\n\r"):
                continue
            if _is_cjk(cp):
                out.append(" " + ch + " ")
            elif ch in " \t\n\r" or unicodedata.category(ch) == "Zs":
                out.append(" ")
            else:
                out.append(ch)
        return "".join(out)

    def basic(self, text):
        """Whitespace tokens of ``text`` with punctuation split off (and lower-cased / accent-stripped)."""
        words = []
        for tok in self._clean(text).split():
            if tok in self.vocab and tok.startswith("[") and tok.endswith("]"):
                words.append(tok)                           # special tokens ([MASK] ...) stay whole
                continue
            if self.lower:
                tok = "".join(c for c in unicodedata.normalize("NFD", tok.lower()) if unicodedata.category(c) != "Mn")
            cur = []
            for ch in tok:
                if _is_punctuation(ch):
                    if cur:
                        words.append("".join(cur))
                        cur = []
                    words.append(ch)
                else:
                    cur.append(ch)
            if cur:
                words.append("".join(cur))
        return words

    # ---- word pieces ------------------------------------------------------------------------------------------------
    def pieces(self, token):
        if len(token) > self.max_chars:
            return [self.unk]
        out, start = [], 0
        while start < len(token):
            end, found = len(token), None
            while start < end:
                sub = token[start:end] if start == 0 else "##" + token[start:end]
                if sub in self.vocab:
                    found = sub
                    break
                end -= 1
            if found is None:
                return [self.unk]
            out.append(found)
            start = end
        return out

    def tokenize(self, text):
        return [p for tok in self.basic(text) for p in self.pieces(tok)]

    def word_ids(self, word):
        """Piece ids of one whitespace-separated word (``tokenizer.tokenize(word)`` in ``_tokenize``)."""
        return tuple(self.vocab[p] for p in self.tokenize(word))

    def words(self, text):
        """``Adv_attack._tokenize`` (adv_attack.py:141-154): ``(words, [piece ids per word])`` of a lower-cased,
        newline-free sentence split at single spaces (an empty word -- two spaces in a row -- has no pieces)."""
        words = text.replace("\n", "").lower().split(" ")
        return words, [self.word_ids(w) for w in words]

    def decode_word(self, ids):
        return "".join(self.tokens[i][2:] if self.tokens[i].startswith("##") else self.tokens[i] for i in ids)
