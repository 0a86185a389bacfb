"""Cases of the white-box encoder fixtures -- shared by the generator (``make_encoder_golden.py``, build container, runs
the REFERENCE's classes) and by ``tests/test_reference_checkpoint.py`` (runs the product).  Pure data + a seeded weight
filler; nothing here touches ``/root/reference``.

A case names a reference-format state dict by (key, shape) listing + seed: the generator lists the keys of the
reference's own modules (``state_dict()`` of ``MultiWayTransformer`` / ``VisionTransformer`` / ``BertEncoder`` ...), fills
them with ``seeded_state_dict`` and records listing + per-tensor checksums in the fixture; a test rebuilds the same state
dict from the listing (same torch build on the GPU box: the CPU generator's stream is identical) and verifies the
checksums before it trusts them.  The tiny cases additionally carry their weights verbatim.
"""
import numpy as np
import torch

# (name, kind, kwargs): sizes of the reference models the generator instantiates
VLMO_CASES = {
    # head size 16 (zero-padded to 64 by _mha.py on the GPU -> the eager block loop + csrc/attn.hip)
    "vlmo_tiny": dict(dim=64, depth=3, heads=4, vlffn_start=2, image_size=32, patch=8, max_text_len=40, vocab=997,
                      n_answers=17, text_lens=(9, 14), seed=101, store_weights=True),
    # head size 64: the graph-free fused encoder (attn.hip + block.hip + library GEMMs), one expert + one VL-FFN layer
    "vlmo_head64": dict(dim=128, depth=2, heads=2, vlffn_start=1, image_size=64, patch=16, max_text_len=40, vocab=997,
                        n_answers=17, text_lens=(7, 12), seed=102, store_weights=False),
    # base width, one expert layer + one VL-FFN layer, 384 px: 14 + 577 = 591 tokens after the padding trim
    "vlmo_base2_384": dict(dim=768, depth=2, heads=12, vlffn_start=1, image_size=384, patch=16, max_text_len=40,
                           vocab=997, n_answers=17, text_lens=(14, 9), seed=103, store_weights=False),
    # the same at 480 px: 14 + 901 = 915 tokens
    "vlmo_base2_480": dict(dim=768, depth=2, heads=12, vlffn_start=1, image_size=480, patch=16, max_text_len=40,
                           vocab=997, n_answers=17, text_lens=(14, 11), seed=104, store_weights=False),
}

ALBEF_CASES = {
    "albef_tiny": dict(dim=64, heads=4, vit_depth=3, bert_depth=3, fusion_layer=1, image_size=32, patch=8, vocab=997,
                       text_len=12, text_lens=(12, 7), seed=201, store_weights=True),
    "albef_head64": dict(dim=128, heads=2, vit_depth=2, bert_depth=2, fusion_layer=1, image_size=64, patch=16, vocab=997,
                         text_len=12, text_lens=(9, 12), seed=202, store_weights=False),
    # base width: two ViT layers at 577 tokens, two BERT layers (the second with cross-attention)
    "albef_base2_384": dict(dim=768, heads=12, vit_depth=2, bert_depth=2, fusion_layer=1, image_size=384, patch=16,
                            vocab=997, text_len=16, text_lens=(16, 10), seed=203, store_weights=False),
}

# The VQA victim (``ALBEF`` of models/model_vqa.py: ViT + fusion BERT + a causal answer decoder with cross-attention in
# every layer, answers RANKED by ``rank_answer``), in the fine-tuned checkpoint's key layout (``text_encoder.*`` without
# ``.bert``, ``text_decoder.bert.*``, ``text_decoder.cls.*``)
ALBEF_VQA_CASES = {
    "albef_vqa_tiny": dict(dim=64, heads=4, vit_depth=2, bert_depth=2, fusion_layer=1, dec_depth=2, image_size=32, patch=8,
                           vocab=997, text_len=12, text_lens=(12, 7), n_answers=13, answer_len=5, k_test=5, seed=301),
    "albef_vqa_head64": dict(dim=128, heads=2, vit_depth=2, bert_depth=2, fusion_layer=1, dec_depth=2, image_size=64,
                             patch=16, vocab=997, text_len=12, text_lens=(9, 12), n_answers=13, answer_len=5, k_test=5,
                             seed=302),
}

SAMPLE_ROWS = 16          # rows of a base-width feature map kept in the fixture
SAMPLE_GRAD = 4096        # elements of a base-width image gradient kept


def seeded_state_dict(listing, seed):
    """{key: tensor} for ``listing`` = [(key, shape, dtype-name)] in the given order.  Deterministic in (listing, seed).

    Rules by key (chosen so that a mix-up in the loader changes results visibly: no tensor is all-zero or all-one):
    LayerNorm weights and layer scales around 1, every bias N(0, 0.1), the relative-position table N(0, 0.5),
    embeddings / class token / position tables N(0, 0.5), matrices N(0, fan_in^-1/2)."""
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for key, shape, dtype in listing:
        shape = tuple(int(s) for s in shape)
        if dtype != "float32":
            if key.endswith("position_ids"):
                out[key] = torch.arange(shape[-1]).expand(shape).clone()
            else:
                out[key] = torch.zeros(shape, dtype=getattr(torch, dtype))
            continue
        leaf = key.split(".")[-1]
        n = torch.empty(shape).normal_(generator=g)
        if "relative_position_bias_table" in key:
            t = 0.5 * n
        elif leaf in ("gamma_1", "gamma_2"):
            t = 0.7 + 0.2 * n
        elif ("norm" in key.lower() and leaf == "weight" and len(shape) == 1):
            t = 1.0 + 0.1 * n
        elif leaf in ("bias", "q_bias", "v_bias") or len(shape) == 1:
            t = 0.1 * n
        elif "embeddings" in key or leaf in ("cls_token", "pos_embed"):
            t = 0.5 * n
        else:
            fan_in = int(np.prod(shape[1:]))
            t = n * fan_in ** -0.5
        out[key] = t.contiguous()
    return out


def listing_of(state_dict):
    return [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in state_dict.items()]


def checksum(t):
    """[sum of the elements' bit patterns, position-weighted sum of them] in wrapping int64 arithmetic: exact, and
    independent of the reduction order (a float sum differs in its last bit between hosts with other thread counts)."""
    if t.dtype != torch.float32:
        bits = t.reshape(-1).to(torch.int64)
    else:
        bits = t.contiguous().reshape(-1).view(torch.int32).to(torch.int64)
    weight = torch.arange(bits.numel(), dtype=torch.int64) % 251 + 1
    return [int(bits.sum()), int((bits * weight).sum())]


def checksums(state_dict):
    return {k: checksum(v) for k, v in state_dict.items()}


def case_inputs(name, case, flavor):
    """Seeded inputs of a case: images (2, 3, H, W) in [-1, 1], token ids / masks (2, L) padded at the end, the MLM copy
    with one position [MASK]-ed (id 103), and the fixed weights of the scalar functional whose gradient is stored."""
    g = torch.Generator().manual_seed(case["seed"] + 5000)
    size, vocab = case["image_size"], case["vocab"]
    length = case["max_text_len"] if flavor == "vlmo" else case["text_len"]
    image = torch.empty(2, 3, size, size).uniform_(-1, 1, generator=g)
    ids = torch.zeros(2, length, dtype=torch.long)
    masks = torch.zeros(2, length, dtype=torch.long)
    for b, n in enumerate(case["text_lens"]):
        ids[b, 0] = 101
        ids[b, 1:n - 1] = torch.randint(200, vocab, (n - 2,), generator=g)
        ids[b, n - 1] = 102
        masks[b, :n] = 1
    mlm_ids = ids.clone()
    mlm_ids[:, 3] = 103
    return dict(image=image, ids=ids, masks=masks, mlm_ids=mlm_ids)


def answer_list(case):
    """Seeded synthetic answer list (n, L) int64: ``[BOS = 1] pieces [SEP = 102] pad = 0`` with 1 .. L - 2 pieces."""
    g = torch.Generator().manual_seed(case["seed"] + 6000)
    n, length = case["n_answers"], case["answer_len"]
    ans = torch.zeros(n, length, dtype=torch.long)
    ans[:, 0] = 1
    for i in range(n):
        k = int(torch.randint(1, length - 1, (1,), generator=g))
        ans[i, 1:1 + k] = torch.randint(200, case["vocab"], (k,), generator=g)
        ans[i, 1 + k] = 102
    return ans


def functional_weights(shapes, seed):
    """One N(0, 1) weight tensor per output: the stored gradient is that of ``sum_i <out_i, W_i>``."""
    g = torch.Generator().manual_seed(int(seed) + 9000)
    return [torch.empty(tuple(s)).normal_(generator=g) for s in shapes]


def sample_rows(n_rows, seed, k=SAMPLE_ROWS):
    g = torch.Generator().manual_seed(int(seed) + 7000)
    k = min(k, n_rows)
    idx = torch.randperm(n_rows, generator=g)[:k].sort().values
    if n_rows > 0 and 0 not in idx.tolist():       # always keep row 0 ([CLS])
        idx[0] = 0
    return idx


def sample_flat(numel, seed, k=SAMPLE_GRAD):
    g = torch.Generator().manual_seed(int(seed) + 8000)
    return torch.randperm(numel, generator=g)[:min(k, numel)].sort().values
