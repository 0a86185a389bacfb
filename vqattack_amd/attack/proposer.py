"""The substitution-candidate proposer of the reference: a text-only BERT with its masked-LM head.

The reference loads ``BertForMaskedLM.from_pretrained('bert-base-uncased')`` (``ALBEF_attack/adv_attack.py:110``;
``vlmo_module.py`` likewise) and, per question, takes the top-5 vocabulary entries of every word position
(``cal_text_attack_list`` :240-244) -- ``propose_candidates`` applies the threshold and the filters to those logits.
Offline there are no weights, so the drivers default to the white box's own MLM head as a stand-in
(``adapters.mlm_logits``); this module is the real thing's architecture for users who have the checkpoint: the HF
``BertForMaskedLM`` state dict loads by its own key names, and the forward runs batched on the device (attention on
``csrc/attn.hip`` through the same ``_BertLayer`` the ALBEF white box uses).  Pinned against the ``transformers``
library's ``BertForMaskedLM`` on seeded weights (``tests/test_proposer.py``).

    proposer = BertMlmProposer.from_hf_state_dict(torch.load("bert-base-uncased/pytorch_model.bin")).to(device)
    attack.attack_mixed(images, ids, masks, attackable, mlm_logits_fn=proposer, ...)
"""
import re

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..whitebox.albef import AlbefConfig, _BertLayer
from ..whitebox.checkpoint import CheckpointError, _copy, _copy_linear, _load_bert_layer, _t, extract_state_dict


class BertMlmProposer(nn.Module):
    def __init__(self, vocab=30522, dim=768, depth=12, heads=12, max_position=512, ln_eps=1e-12):
        super().__init__()
        cfg = AlbefConfig(dim=dim, heads=heads, bert_ln_eps=ln_eps)
        self.word_embeddings = nn.Embedding(vocab, dim)
        self.position_embeddings = nn.Embedding(max_position, dim)
        self.type_embeddings = nn.Embedding(2, dim)
        self.emb_ln = nn.LayerNorm(dim, eps=ln_eps)
        self.layers = nn.ModuleList([_BertLayer(cfg, False) for _ in range(depth)])
        self.head_dense = nn.Linear(dim, dim)
        self.head_ln = nn.LayerNorm(dim, eps=ln_eps)
        self.head_bias = nn.Parameter(torch.zeros(vocab))
        self.eval()
        for p in self.parameters():
            p.requires_grad_(False)

    @classmethod
    def from_hf_state_dict(cls, state_dict, heads=None):
        """Build for and fill from a ``BertForMaskedLM`` state dict (``bert.embeddings.*``, ``bert.encoder.layer.N.*``,
        ``cls.predictions.*``; the decoder is tied to the word embeddings)."""
        sd = extract_state_dict(state_dict)
        word = _t(sd, "bert.embeddings.word_embeddings.weight")
        layers = [int(m.group(1)) for k in sd for m in [re.match(r"bert\.encoder\.layer\.(\d+)\.", k)] if m]
        if not layers:
            raise CheckpointError("no bert.encoder.layer.* keys: not a BertForMaskedLM state dict")
        dim = int(word.shape[1])
        model = cls(vocab=int(word.shape[0]), dim=dim, depth=max(layers) + 1, heads=heads or max(1, dim // 64),
                    max_position=int(_t(sd, "bert.embeddings.position_embeddings.weight").shape[0]))
        model.load_hf_state_dict(sd)
        return model

    def load_hf_state_dict(self, state_dict):
        sd = extract_state_dict(state_dict)
        used = set()
        e = "bert.embeddings."
        _copy(self.word_embeddings.weight, _t(sd, e + "word_embeddings.weight"), e + "word_embeddings.weight")
        _copy(self.position_embeddings.weight, _t(sd, e + "position_embeddings.weight"), e + "position_embeddings.weight")
        _copy(self.type_embeddings.weight, _t(sd, e + "token_type_embeddings.weight"), e + "token_type_embeddings.weight")
        _copy_linear(self.emb_ln, sd, e + "LayerNorm", used)
        for i, layer in enumerate(self.layers):
            _load_bert_layer(layer, sd, "bert.encoder.layer.{}.".format(i), used, self.word_embeddings.weight.shape[1])
        _copy_linear(self.head_dense, sd, "cls.predictions.transform.dense", used)
        _copy_linear(self.head_ln, sd, "cls.predictions.transform.LayerNorm", used)
        _copy(self.head_bias, _t(sd, "cls.predictions.bias"), "cls.predictions.bias")
        return self

    @torch.no_grad()
    def forward(self, text_ids, text_masks=None):
        """MLM logits (B, L, V) for token ids (B, L); ``text_masks`` (B, L) 1 = token, 0 = padding (None: no padding,
        the reference's single-sentence call)."""
        length = text_ids.shape[1]
        x = self.word_embeddings(text_ids) + self.type_embeddings.weight[0]
        x = self.emb_ln(x + self.position_embeddings.weight[:length].unsqueeze(0))
        pad = None
        if text_masks is not None:
            pad = torch.zeros(text_masks.shape[0], 1, 1, length, device=x.device)
            pad = pad.masked_fill(~text_masks.bool()[:, None, None, :], float("-inf"))
        for layer in self.layers:
            x = layer(x, pad, None)
        h = self.head_ln(F.gelu(self.head_dense(x)))
        return F.linear(h, self.word_embeddings.weight, self.head_bias)


def banned_ids(vocab_tokens, stop_words=()):
    """bool (V,): ids ``propose_candidates`` may never propose -- ``##`` word pieces ("filter out sub-word") and stop words
    (``filter_words``), ``adv_attack.py:253-258``."""
    stop = set(stop_words)
    return torch.tensor([t.startswith("##") or t in stop for t in vocab_tokens], dtype=torch.bool)
