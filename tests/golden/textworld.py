"""The synthetic text world of the text-side golden fixtures: vocabulary, tokenizer, embedding tables, stand-ins.

Shared by the generator (``make_text_golden.py``, build container) and by the tests that replay the fixtures, so both
sides build the SAME tokenizer / tables from the data stored in the fixture.  Nothing here touches the reference.

  * vocabulary: BERT-uncased layout for the special ids the reference hard-codes ([PAD] 0, [UNK] 100, [CLS] 101,
    [SEP] 102, [MASK] 103; ``vlmo_module.py:1908`` searches for the literal 102), then whole words, stop words and
    ``##`` word pieces;
  * tokenizer: Hugging Face ``BertTokenizer`` (transformers, the third-party WordPiece implementation the reference
    uses through ``BertTokenizer.from_pretrained('bert-base-uncased')``, ``adv_attack.py:51-52``) over that vocabulary;
  * sentence encoder stand-in for TF-Hub USE (``adv_attack.py:101-103,315-318``): mean of a fixed random table over the
    sentence's token ids (special tokens included), i.e. exactly ``BagOfEmbeddingsSimilarity`` on token ids;
  * MLM stand-in for ``BertForMaskedLM`` (``adv_attack.py:110``): a fixed random table lookup ``ids -> logits``.
"""
import os
import tempfile

import numpy as np
import torch

PAD, UNK, CLS, SEP, MASK = 0, 100, 101, 102, 103

WORDS = ("what color is the cat dog sitting on table how many people are in picture man woman holding red blue "
         "green umbrella kite ball playing frisbee field grass beach water standing near bus train street sign "
         "large small white black brown yellow horse zebra giraffe elephant pizza plate food eating kitchen room "
         "bed couch chair wearing hat shirt glasses two three four one snow ski board wall clock tower building "
         "sky tree flying plane bird boat river bridge old young tall stone wooden metal open closed left right "
         "which animal sport vehicle fruit banana apple orange bowl cup bottle laptop phone book desk window "
         "and his her its a of there sees see other animals does where dark 2").split()
PIECES = ["##s", "##ing", "##ed", "##er", "##ly", "##x", "##y"]
PUNCT = ["?", ".", ",", "'"]


def build_vocab():
    vocab = ["[PAD]"] + ["[unused{}]".format(i) for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    seen = set(vocab)
    for w in PUNCT + WORDS + PIECES:
        if w not in seen:
            seen.add(w)
            vocab.append(w)
    return vocab


def make_tokenizer(vocab):
    """HF BertTokenizer over ``vocab`` (list of tokens, index = id)."""
    from transformers import BertTokenizer
    d = tempfile.mkdtemp(prefix="vqa_vocab_")
    path = os.path.join(d, "vocab.txt")
    with open(path, "w") as fh:
        fh.write("\n".join(vocab) + "\n")
    tok = BertTokenizer(path, do_lower_case=True)
    assert tok.pad_token_id == PAD and tok.cls_token_id == CLS and tok.sep_token_id == SEP and tok.mask_token_id == MASK
    return tok


def seeded(seed, *shape, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def embedding_tables(vocab_size, dim, max_pos=64, seed=7):
    """BERT embedding tables (word, position, type, LayerNorm gamma / beta) with non-trivial LayerNorm parameters."""
    return dict(word=seeded(seed, vocab_size, dim, scale=0.5), pos=seeded(seed + 1, max_pos, dim, scale=0.5),
                type_emb=seeded(seed + 2, 2, dim, scale=0.5),
                gamma=(1.0 + 0.1 * seeded(seed + 3, dim)).astype(np.float32), beta=seeded(seed + 4, dim, scale=0.1),
                ln_eps=1e-12)


class SentenceEncoderStandIn:
    """``USE_model([a, b]).numpy()`` stand-in: one row per sentence = mean of ``table[id]`` over the sentence's ids as the
    tokenizer encodes it (with [CLS] / [SEP], without padding)."""

    def __init__(self, tokenizer, table):
        self.tok, self.table = tokenizer, table

    def embed_ids(self, ids):
        ids = [int(t) for t in ids if int(t) != PAD]
        return self.table[ids].mean(0)

    def __call__(self, sentences):
        rows = [self.embed_ids(self.tok(s)["input_ids"]) for s in sentences]
        out = np.stack(rows).astype(np.float32)

        class _T:
            def numpy(self_inner):
                return out
        return _T()

    def similarity_ids(self, ori_ids, new_ids):
        """The reference's USE_sim arithmetic (adv_attack.py:316-318) on two id rows."""
        embs = np.stack([self.embed_ids(ori_ids), self.embed_ids(new_ids)]).astype(np.float32)
        norm = np.linalg.norm(embs, axis=1)
        embs = embs / norm[:, None]
        return (embs[:1] * embs[1:]).sum(axis=1)[0]


class MlmStandIn:
    """``mlm_model(input_ids)[0]`` stand-in: logits (1, L, V) = table[id] + position drift, plus hand-placed boosts so
    that every branch of ``cal_text_attack_list`` / ``get_substitues`` is exercised (original word in the top-5, ``##``
    pieces, stop words, scores below the 0.3 threshold)."""

    def __init__(self, table, drift):
        self.table, self.drift = table, drift      # (V, V) and (P, V)

    def logits_for(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        return (self.table[ids] + self.drift[:len(ids)]).astype(np.float32)

    def __call__(self, input_ids):
        return (torch.from_numpy(np.stack([self.logits_for(row.tolist()) for row in input_ids])),)

    def to(self, *_a, **_k):
        return self
