"""Frozen VLMo-shaped white box (multiway transformer) and its batched attack adapters -- PyTorch-ROCm plumbing.

Architecture follows the reference's ``vlmo/modules/multiway_transformer.py`` (``Attention`` :58-118, ``Block`` :121-201,
``visual_embed`` :366-380, base/large factories :385-404) and the embedding / head wiring of
``vlmo/modules/vlmo_module.py`` (``pgd_attack`` :1387-1446, ``pgd_attack_vl`` :1328-1385, ``pgd_mlm_attack`` :1448-1529,
``vqa_classifier`` :274-279).  Weights are randomly initialised from a seed (no checkpoints exist offline) and frozen;
only the *shape* of the computation matters for the attack path and the metric.

Differences from the reference that serve the MI355X design (none changes a single-sample result):
  * adapters are BATCHED: the reference hard-codes batch 1 (``target_feats[0, ...]``); here per-layer features stay
    ``(B, T+N, D)`` and padded text tokens are excluded with a row-weight mask instead of a ragged gather;
  * per-layer features are returned as ``LayerFeatures`` (no ``torch.stack`` copy);
  * the patch embedding is an unfold + GEMM (``F.linear``) instead of a strided ``Conv2d``: its backward w.r.t. the
    image -- the producer of the gradient the fused step kernel consumes -- is then a plain GEMM;
  * attention is the exact-fp32 MFMA kernel of ``csrc/attn.hip`` (``_mha.py``) on the packed qkv projection, with the
    relative-position bias and key-padding mask folded into one additive mask.
"""
import os
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..features import LayerFeatures
from . import _fused
from ._mha import mha_packed


@dataclass
class VlmoConfig:
    dim: int = 768
    depth: int = 12
    heads: int = 12
    vlffn_start: int = 10
    mlp_ratio: float = 4.0
    patch: int = 16
    image_size: int = 384
    max_text_len: int = 40
    vocab: int = 30522
    max_position: int = 512
    layer_scale: float = 0.1
    n_answers: int = 3129
    ln_eps: float = 1e-6
    bert_ln_eps: float = 1e-12
    # Synthetic weights (no checkpoints offline).  "trained_like": unit-gain linear layers (std = fan_in^-1/2) and layer
    # scale 1, so activations stay O(1) through the depth and predictions depend on the image, as in a trained network.
    # "pretrain_init": the reference's initialisation before any training (trunc-normal 0.02, layer scale 0.1) -- with
    # it every output is dominated by the token embeddings and no perturbation ever changes a black-box answer.
    weights: str = "trained_like"
    # BertEmbeddings adds its absolute position table only for position_embedding_type "absolute"; the reference builds
    # VLMo-base / -large with "rel_pos" (vlmo_module.py:216; transformers==4.8.1 BertEmbeddings, the code vendored at
    # ALBEF_attack/models/xbert.py:210-212), so a reference checkpoint runs with False (whitebox/checkpoint.py sets it).
    # True is the synthetic default the committed fixtures and recorded numbers were made with.
    text_abs_pos: bool = True

    @property
    def n_patches(self):
        return (self.image_size // self.patch) ** 2

    @property
    def n_image_tokens(self):
        return self.n_patches + 1


def vlmo_base(image_size=384, **kw):
    return VlmoConfig(dim=768, depth=12, heads=12, vlffn_start=10, image_size=image_size, **kw)


def vlmo_large(image_size=384, **kw):
    return VlmoConfig(dim=1024, depth=24, heads=16, vlffn_start=21, image_size=image_size, **kw)


def vlmo_tiny(**kw):   # test-sized
    return VlmoConfig(dim=64, depth=3, heads=4, vlffn_start=2, image_size=32, patch=8, max_text_len=8, vocab=30522,
                      n_answers=17, **kw)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)   # reference: q_bias/v_bias parameters, zero k bias
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, bias):
        b, n, c = x.shape
        o = mha_packed(self.qkv(x).reshape(b, n, 3, self.heads, c // self.heads), bias)
        return self.proj(o.reshape(b, n, c))


class _SplitTokens(torch.autograd.Function):
    """``x[:, :n], x[:, n:]`` as two contiguous tensors (what LayerNorm / Linear would copy them into anyway), with a
    backward that writes the two gradients into one buffer in a single pass -- autograd's own slice backward allocates a
    zero-filled full-size tensor per slice, copies the slice gradient into it and adds the two (5 passes over the
    activations per layer instead of 1)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.shape = n, x.shape
        return x[:, :n].contiguous(), x[:, n:].contiguous()

    @staticmethod
    def backward(ctx, gt, gi):
        b, s, d = ctx.shape
        if gt is None:
            gt = gi.new_zeros(b, ctx.n, d)
        if gi is None:
            gi = gt.new_zeros(b, s - ctx.n, d)
        return torch.cat([gt, gi], dim=1), None


class Block(nn.Module):
    def __init__(self, cfg, with_vlffn):
        super().__init__()
        hidden = int(cfg.dim * cfg.mlp_ratio)
        self.norm1 = nn.LayerNorm(cfg.dim, eps=cfg.ln_eps)
        self.attn = Attention(cfg.dim, cfg.heads)
        self.norm2_text = nn.LayerNorm(cfg.dim, eps=cfg.ln_eps)
        self.norm2_imag = nn.LayerNorm(cfg.dim, eps=cfg.ln_eps)
        self.mlp_text = Mlp(cfg.dim, hidden)
        self.mlp_imag = Mlp(cfg.dim, hidden)
        self.mlp_vl = Mlp(cfg.dim, hidden) if with_vlffn else None
        self.norm2_vl = nn.LayerNorm(cfg.dim, eps=cfg.ln_eps) if with_vlffn else None
        self.gamma_1 = nn.Parameter(cfg.layer_scale * torch.ones(cfg.dim))
        self.gamma_2 = nn.Parameter(cfg.layer_scale * torch.ones(cfg.dim))
        self.max_text_len = cfg.max_text_len

    def forward(self, x, bias, text_len=None):
        x = torch.addcmul(x, self.gamma_1, self.attn(self.norm1(x), bias))
        if self.mlp_vl is None:     # modality experts: text tokens / image tokens (multiway_transformer.py:193-197)
            n_text = self.max_text_len if text_len is None else text_len
            t, i = _SplitTokens.apply(x, n_text)
            t = torch.addcmul(t, self.gamma_2, self.mlp_text(self.norm2_text(t)))
            i = torch.addcmul(i, self.gamma_2, self.mlp_imag(self.norm2_imag(i)))
            return torch.cat([t, i], dim=1)
        return torch.addcmul(x, self.gamma_2, self.mlp_vl(self.norm2_vl(x)))


class FrozenVlmo(nn.Module):
    """White box (pre-trained role) and, with ``vqa_head=True``, the black-box VQA scorer (fine-tuned role)."""

    def __init__(self, cfg, seed=0, vqa_head=False):
        super().__init__()
        self.cfg = cfg
        d = cfg.dim
        self.word_embeddings = nn.Embedding(cfg.vocab, d)
        self.position_embeddings = nn.Embedding(cfg.max_position, d)
        self.bert_type_embeddings = nn.Embedding(2, d)
        self.bert_ln = nn.LayerNorm(d, eps=cfg.bert_ln_eps)
        self.token_type_embeddings = nn.Embedding(2, d)
        self.patch_proj = nn.Linear(3 * cfg.patch * cfg.patch, d)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, d))
        self.blocks = nn.ModuleList([Block(cfg, i >= cfg.vlffn_start) for i in range(cfg.depth)])
        self.norm = nn.LayerNorm(d, eps=cfg.ln_eps)
        self.pooler = nn.Linear(d, d)
        self.mlm_dense = nn.Linear(d, d)
        self.mlm_ln = nn.LayerNorm(d, eps=cfg.bert_ln_eps)
        self.mlm_bias = nn.Parameter(torch.zeros(cfg.vocab))
        # MLM decoder: tied to the word embeddings for the synthetic weights; the reference's ``MLMHead(bert_config)``
        # (vlmo_module.py:232, heads.py:40-47) has its OWN (V, D) matrix, which ``set_mlm_decoder`` installs
        self.register_parameter("mlm_decoder", None)
        ntok = cfg.max_text_len + cfg.n_image_tokens
        # relative position bias, one (heads, T+N, T+N) slab per layer (vlmo_module.py:807-814); frozen -> precomputed
        self.register_buffer("rel_pos_bias", torch.zeros(cfg.depth, cfg.heads, ntok, ntok), persistent=False)
        self.vqa_classifier = None
        if vqa_head:
            self.vqa_classifier = nn.Sequential(nn.Linear(d, 2 * d), nn.LayerNorm(2 * d), nn.GELU(),
                                                nn.Linear(2 * d, cfg.n_answers))
        self._init(seed)
        self.eval()
        for p in self.parameters():
            p.requires_grad_(False)
        # On the GPU the block loop runs without an autograd graph inside (library GEMMs + csrc/attn.hip + the fused
        # block glue of csrc/block.hip, whitebox/_fused.py) when the shapes are the hand-written kernels' (head size 64:
        # every BASELINE configuration); False keeps the eager nn.Module loop (parity tests compare the two).
        self.fused_blocks = os.environ.get("VQA_FUSED_BLOCKS", "1") != "0"      # "0": A/B measurements of the eager loop
        self._fused_spec = None

    def _apply(self, fn, *args, **kwargs):
        self._fused_spec = None                  # .to(device): the spec holds device tensors of the frozen weights
        return super()._apply(fn, *args, **kwargs)

    def _init(self, seed):
        g = torch.Generator().manual_seed(seed)

        def normal_(t):
            t.copy_(torch.empty(t.shape).normal_(0.0, 0.02, generator=g))

        trained = self.cfg.weights == "trained_like"
        with torch.no_grad():
            for mod in self.modules():
                if isinstance(mod, nn.Linear):
                    if trained:
                        mod.weight.copy_(torch.empty(mod.weight.shape).normal_(0.0, mod.in_features ** -0.5, generator=g))
                    else:
                        normal_(mod.weight)
                    if mod.bias is not None:
                        mod.bias.zero_()
                elif isinstance(mod, nn.Embedding):
                    normal_(mod.weight)
                elif isinstance(mod, nn.LayerNorm):
                    mod.weight.fill_(1.0)
                    mod.bias.zero_()
            if trained:
                for blk in self.blocks:
                    blk.gamma_1.fill_(1.0)
                    blk.gamma_2.fill_(1.0)
            normal_(self.cls_token)
            normal_(self.mlm_bias)
            normal_(self.rel_pos_bias)

    @classmethod
    def finetuned_from(cls, white, seed=1, drift=0.25):
        """Synthetic black box: the reference's VQA victim is the pre-trained white box after fine-tuning
        (``reload_vqa`` / ``reload_pretrain`` swap two checkpoints of ONE architecture, vlmo_module.py:330-567), so its
        trunk is a perturbed copy of the white box's -- every weight moves by ``drift`` of its own RMS -- plus a
        freshly initialised answer head."""
        black = cls(white.cfg, seed=seed, vqa_head=True)
        g = torch.Generator().manual_seed(seed + 7919)
        src = dict(white.named_parameters())
        with torch.no_grad():
            for name, p in black.named_parameters():
                if name in src:
                    w = src[name].detach().cpu()
                    rms = float(w.pow(2).mean().sqrt())
                    p.copy_(w + drift * rms * torch.empty(w.shape).normal_(generator=g))
            black.rel_pos_bias.copy_(white.rel_pos_bias.detach().cpu())
        return black

    def invalidate_fused(self):
        """Drop the cached fused-encoder spec: the next pass rebuilds it from the current weights.  Needed only after
        weight updates ``_fused.weights_key`` cannot see (writes through ``p.data``)."""
        self._fused_spec = None

    # ---- reference checkpoints ------------------------------------------------------------------------------
    def set_mlm_decoder(self, weight):
        """Untie the MLM decoder from the word embeddings (``weight`` (V, D)); None ties it again."""
        if weight is None:
            self.mlm_decoder = None
            return
        ref = self.word_embeddings.weight
        self.mlm_decoder = nn.Parameter(weight.detach().to(ref.device, ref.dtype).clone(), requires_grad=False)

    def load_reference_state_dict(self, state_dict, strict=True, max_text_len_of_initckpt=None):
        """Fill this model from a state dict of the reference's ``VLMo`` module (``transformer.blocks.N.attn.q_bias``,
        ``relative_position_bias_table``, ``text_embeddings.*``, ...): ``whitebox/checkpoint.py``.  The configuration
        must already have the checkpoint's widths (``checkpoint.vlmo_from_reference`` derives it); the text embeddings
        switch to the reference's "rel_pos" form (no absolute position table added)."""
        from . import checkpoint
        out = checkpoint.load_vlmo(self, state_dict, strict=strict, max_text_len_of_initckpt=max_text_len_of_initckpt)
        self.cfg.text_abs_pos = False
        self._zero_pos = None
        return out

    # ---- embeddings -----------------------------------------------------------------------------------------
    def text_embeddings(self, ids):
        """BERT embeddings (word + type 0, then + position when ``cfg.text_abs_pos``, LayerNorm); eval mode -> no
        dropout."""
        length = ids.shape[1]
        e = self.word_embeddings(ids) + self.bert_type_embeddings.weight[0]
        if self.cfg.text_abs_pos:
            e = e + self.position_embeddings.weight[:length].unsqueeze(0)
        return self.bert_ln(e)

    def embedding_tables(self):
        """Raw tables for the candidate-scoring kernel (``ops.cand_dir_sim``)."""
        pos = self.position_embeddings.weight
        if not self.cfg.text_abs_pos:                    # the kernels add a position row: hand them exact zeros
            z = getattr(self, "_zero_pos", None)
            if z is None or z.device != pos.device or z.shape != pos.shape:
                z = self._zero_pos = torch.zeros_like(pos)
            pos = z
        return dict(word=self.word_embeddings.weight, pos=pos,
                    type_emb=self.bert_type_embeddings.weight, gamma=self.bert_ln.weight, beta=self.bert_ln.bias,
                    ln_eps=self.cfg.bert_ln_eps)

    def visual_embed(self, image):
        """``image``: (B, 3, H, W), or already patch-major (B, n_patches, 3*p*p) (``vqattack_amd.layout``)."""
        b, p = image.shape[0], self.cfg.patch
        g = self.cfg.image_size // p
        patches = image if image.dim() == 3 else \
            image.reshape(b, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(b, g * g, 3 * p * p)
        x = self.patch_proj(patches)
        return torch.cat([self.cls_token.expand(b, -1, -1), x], dim=1)

    # ---- trunk ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def attention_bias(self, text_masks):
        """Per-layer additive attention masks = relative-position bias + key padding (-inf on padded text keys).

        ``text_masks`` may be shorter than ``max_text_len`` (trailing padding trimmed by the adapter): the bias is then
        gathered for the kept positions only -- a masked key contributes exp(-inf) = 0 to every softmax, so dropping
        it changes no result.  The masks depend only on the text batch, so an attack builds them ONCE per text batch
        and reuses them for every PGD step.  Layout choices that keep the fused attention kernel off the HBM roof:
          * rows are padded to a multiple of 32 floats in storage and sliced back: the attention kernels read a row
            in whole 32-key tiles of aligned 16-byte groups, and neither they nor the library re-pad (= copy) the
            mask on every call;
          * when every question of the batch has the same padding pattern (batches are bucketed by schedule and
            length) the mask is ONE (1, heads, S, S) slab expanded over the batch with stride 0: 18 MB per layer that
            stays cache-resident instead of a (B, heads, S, S) tensor (1.17 GB at batch 64) streamed by every
            forward attention kernel;
          * a RAGGED batch on the GPU (questions of different lengths, padded at the end -- what a tokenizer produces)
            keeps that one slab too: the padded text keys of sample b are the range [length_b, n_text), handed to the
            attention kernel as a per-sample "key hole" (``attention.KeyHoleBias``) instead of being written into a
            per-sample copy of the slab.  Only masks that are not of that form (or host tensors: the CPU oracle's
            model) are materialised per sample.
        """
        from ..attention import KeyHoleBias
        b, n_text = text_masks.shape
        dev = text_masks.device
        s = n_text + self.cfg.n_image_tokens
        s_pad = (s + 31) // 32 * 32
        keep = torch.cat([text_masks.bool(), torch.ones(b, self.cfg.n_image_tokens, dtype=torch.bool, device=dev)], dim=1)
        shared = bool((keep == keep[:1]).all())          # one host sync per text batch
        hole = None
        if not shared and dev.type == "cuda":
            lengths = text_masks.bool().sum(dim=1)
            prefix = text_masks.bool() == (torch.arange(n_text, device=dev)[None, :] < lengths[:, None])
            if bool(prefix.all()):
                hole = torch.stack([lengths, torch.full_like(lengths, n_text)], dim=1).to(torch.int32).contiguous()
                shared = True                            # the slab carries the relative-position bias only
                keep = torch.ones_like(keep)
        rows = keep[:1] if shared else keep
        pad = torch.zeros(rows.shape[0], 1, 1, s, device=dev).masked_fill(~rows[:, None, None, :], float("-inf"))
        if n_text == self.cfg.max_text_len:
            rel = self.rel_pos_bias
        else:                                            # positions [0, n_text) + the image block
            idx = torch.cat([torch.arange(n_text, device=dev),
                             torch.arange(self.cfg.max_text_len, self.cfg.max_text_len + self.cfg.n_image_tokens,
                                          device=dev)])
            rel = self.rel_pos_bias[:, :, idx][:, :, :, idx]
        out = []
        for li in range(self.cfg.depth):
            store = torch.zeros(rows.shape[0], self.cfg.heads, s, s_pad, device=dev)
            store[..., :s] = pad + rel[li].unsqueeze(0)
            view = store[..., :s]
            view = view.expand(b, -1, -1, -1) if shared else view
            out.append(view if hole is None else KeyHoleBias(view, hole))
        return out

    def encode(self, image, text_embeds, text_masks, bias=None):
        """Returns (per-layer inputs/outputs list of depth+1 tensors (B, T+N, D), final normed states); T is the text
        length actually passed (<= max_text_len)."""
        n_text = text_embeds.shape[1]
        t = text_embeds + self.token_type_embeddings.weight[0]
        i = self.visual_embed(image) + self.token_type_embeddings.weight[1]
        x = torch.cat([t, i], dim=1)
        if bias is None:
            bias = self.attention_bias(text_masks)
        if self.fused_blocks and x.is_cuda and x.dtype == torch.float32 and _fused.supported(self.cfg.dim, self.cfg.heads):
            key = (_fused.weights_key(self.blocks), _fused.weights_key(self.norm))
            if self._fused_spec is None or self._fused_spec[0] != key:      # load_state_dict / in-place weight updates
                self._fused_spec = (key, _fused.vlmo_spec(self))
            return _fused.encode(x, self._fused_spec[1], bias, n_text)
        feats = [x]
        for li, blk in enumerate(self.blocks):
            x = blk(x, bias[li], n_text)
            feats.append(x)
        return feats, self.norm(x)

    def mlm_score(self, text_states):
        h = self.mlm_ln(F.gelu(self.mlm_dense(text_states)))
        return F.linear(h, self.word_embeddings.weight if self.mlm_decoder is None else self.mlm_decoder, self.mlm_bias)

    def pooled(self, states):
        return torch.tanh(self.pooler(states[:, 0]))

    @torch.no_grad()
    def vqa_answer(self, image, text_ids, text_masks, n_answers=None):
        """Black-box prediction: argmax over the answer vocabulary (objectives.vqa_test_step_after_pgd, :812-829).
        ``n_answers``: restrict the victim to the first ``n_answers`` classes (a question type with a closed answer set,
        e.g. yes / no); None = the whole vocabulary."""
        if self.vqa_classifier is None:
            raise RuntimeError("this FrozenVlmo was built without a VQA head")
        _, states = self.encode(image, self.text_embeddings(text_ids), text_masks)
        return self.vqa_classifier(self.pooled(states))[:, :n_answers].argmax(dim=-1)


class VlmoAttackAdapters:
    """Batched ``model_fn`` closures over the current text batch (the reference's ``self.batch``).

    ``trim_padding``: questions are padded to ``max_text_len`` = 40 but are 6-14 tokens long; when the batch's masks are
    prefix masks the trailing all-padding columns are not run through the encoder at all (padded keys are masked to
    -inf and padded query rows carry no loss, so every result is unchanged; ~5 % fewer tokens, ~10 % less attention).
    """

    def __init__(self, model, trim_padding=True):
        self.model = model
        self.batch = {}
        self.trim_padding = trim_padding
        self._tlen = model.cfg.max_text_len
        self._mlm_rows = None
        self._mlm_samples = None

    def _text_len(self, masks):
        full = self.model.cfg.max_text_len
        if not self.trim_padding or masks.shape[1] != full:
            return masks.shape[1]
        m = masks.bool()
        lengths = m.sum(dim=1)
        prefix = bool((m == (torch.arange(full, device=m.device)[None, :] < lengths[:, None])).all())
        return max(int(lengths.max()), 1) if prefix else full       # one host sync per text batch

    def set_text(self, text_ids, text_masks, text_ids_mlm=None, text_mask_mlm=None, text_len=None):
        """``text_len`` pins the trimmed text length (callers that change the batch between ``gen_ori_feats`` and the
        attack steps must keep the token layout of the targets)."""
        n = text_len or self._text_len(text_masks if text_mask_mlm is None else (text_masks | text_mask_mlm))
        self._tlen = n
        self.batch["text_ids"], self.batch["text_masks"] = text_ids[:, :n], text_masks[:, :n]
        self.batch["text_ids_mlm"] = self.batch["text_ids"] if text_ids_mlm is None else text_ids_mlm[:, :n]
        self.batch["text_mask_mlm"] = self.batch["text_masks"] if text_mask_mlm is None else text_mask_mlm[:, :n]
        self._weight = None
        self._bias = self.model.attention_bias(self.batch["text_masks"])
        self._bias_mlm = self._bias if text_mask_mlm is None else self.model.attention_bias(self.batch["text_mask_mlm"])

    def set_mlm_rows(self, rows):
        """Live-rows form of the MLM closure: ``rows`` int64 (B, W) = the text positions whose MLM labels are targets
        (``mlm_task.live_label_rows``); ``pgd_mlm_attack`` then returns logits (B, W, V) for those positions only, to be
        scored against the equally compacted labels.  ``None`` restores the dense (B, max_text_len, V) contract of the
        reference's closure (vlmo_module.py:1448-1529).  In the reference's workload >= 90 % of the positions carry
        ``ignore_index``: the dense form runs the 768 x 30522 head on them, writes their logits, and the loss writes
        their all-zero gradient rows (625 MB each way per iteration at batch 64)."""
        self._mlm_rows = None if rows is None else rows.contiguous()

    def set_mlm_samples(self, index):
        """int64 device tensor with the batch indices of the samples that take an MLM step in the following
        ``pgd_attack_mixed`` calls (None / empty: nobody does)."""
        self._mlm_samples = None if index is None or index.numel() == 0 else index

    def save_text(self):
        """Opaque snapshot of the current text batch (ids, masks, trimmed length, row weights, attention masks): a
        driver that alternates between two text batches restores them with ``load_text`` instead of rebuilding the
        per-layer attention masks every step."""
        return (dict(self.batch), self._tlen, self._weight, self._bias, self._bias_mlm)

    def load_text(self, state):
        batch, self._tlen, self._weight, self._bias, self._bias_mlm = state
        self.batch = dict(batch)

    def text_embeddings(self, ids):
        return self.model.text_embeddings(ids)

    def mlm_logits(self, text_ids, text_masks):
        """Candidate proposer stand-in for the reference's separate HF BERT-MLM (adv_attack.py:110,242): the white
        box's own MLM head over a blank image."""
        m = self.model
        with torch.no_grad():
            zeros = torch.zeros(text_ids.shape[0], 3, m.cfg.image_size, m.cfg.image_size, device=text_ids.device)
            _, states = m.encode(zeros, m.text_embeddings(text_ids), text_masks)
            return m.mlm_score(states[:, :text_ids.shape[1]])

    def row_weight(self):
        """uint8 (B, T+N): 2 for the [CLS] row (counted alone and as a token by the VLMO loss), 1 for real text tokens
        and all image tokens, 0 for padded text tokens."""
        if self._weight is None:
            m = self.batch["text_masks"]
            w = torch.cat([m.to(torch.uint8), torch.ones(m.shape[0], self.model.cfg.n_image_tokens, dtype=torch.uint8,
                                                         device=m.device)], dim=1)
            w[:, 0] = 2
            self._weight = w.contiguous()
        return self._weight

    def _pack(self, feats, states):
        return [self.model.pooled(states), None, LayerFeatures(feats, self.row_weight())]

    def gen_ori_feats(self, image):
        """Targets of the feature loss from the clean pair (``Gen_ori_feats``, vlmo_module.py:1287-1312)."""
        with torch.no_grad():
            m = self.model
            feats, states = m.encode(image, m.text_embeddings(self.batch["text_ids"]), self.batch["text_masks"],
                                     self._bias)
        return [m.pooled(states), None, LayerFeatures(feats, self.row_weight())]

    def pgd_attack(self, x):
        m = self.model
        feats, states = m.encode(x, m.text_embeddings(self.batch["text_ids"]), self.batch["text_masks"], self._bias)
        return self._pack(feats, states)

    def pgd_attack_vl(self, xs):
        # xs[1] holds the embeddings of all max_text_len positions; the trimmed columns simply get a zero gradient
        feats, states = self.model.encode(xs[0], xs[1][:, :self._tlen], self.batch["text_masks"], self._bias)
        return self._pack(feats, states)

    def pgd_attack_mixed(self, xs):
        """One encoder pass for a batch whose samples stand at different steps of their schedules (``attack_mixed``):
        the current text batch holds, per sample, the question (feature step: ``pgd_attack`` / ``pgd_attack_vl``,
        vlmo_module.py:1387-1446 / :1328-1385) or the [MASK]-ed paraphrase (MLM step: ``pgd_mlm_attack``, :1448-1529).
        Returns ``(features, logits)``: the feature list with the rows of the MLM-step samples weighted 0 (None when every
        sample is one), and the MLM logits (n_mlm, W, V) of those samples at their live label rows (None when there are
        none).  ``xs[1]`` holds the text embeddings of all positions, per sample those of the text it is run with."""
        m = self.model
        feats, states = m.encode(xs[0], xs[1][:, :self._tlen], self.batch["text_masks"], self._bias)
        sel = self._mlm_samples
        if sel is None:
            return self._pack(feats, states), None
        if self._mlm_rows is None:
            raise RuntimeError("pgd_attack_mixed with MLM-step samples needs set_mlm_rows() first")
        idx = self._mlm_rows[sel].unsqueeze(-1).expand(-1, -1, states.shape[-1])
        logits = m.mlm_score(torch.gather(states[sel], 1, idx))
        if sel.numel() == xs[0].shape[0]:
            return None, logits
        w = self.row_weight().clone()
        w[sel] = 0
        return [m.pooled(states), None, LayerFeatures(feats, w)], logits

    def pgd_mlm_attack(self, x):
        m = self.model
        feats, states = m.encode(x, m.text_embeddings(self.batch["text_ids_mlm"]), self.batch["text_mask_mlm"],
                                 self._bias_mlm)
        if self._mlm_rows is not None:          # live rows only: no logits for positions whose labels are ignore_index
            idx = self._mlm_rows.unsqueeze(-1).expand(-1, -1, states.shape[-1])
            return [m.mlm_score(torch.gather(states, 1, idx)), None, LayerFeatures(feats, self.row_weight())]
        logits = m.mlm_score(states[:, :self._tlen])
        if self._tlen < m.cfg.max_text_len:     # dense contract: labels cover max_text_len positions (trimmed ones must
            logits = F.pad(logits, (0, 0, 0, m.cfg.max_text_len - self._tlen))   # be ignore_index; the runner checks)
        return [logits, None, LayerFeatures(feats, self.row_weight())]
