"""Tiny deterministic white-box stand-ins used by the golden fixtures and parity tests.

They follow the ``model_fn`` contract of the reference's attack operators
(``x -> list[Tensor]``, differentiable w.r.t. ``x``) at toy sizes so the
CPU oracle finishes in milliseconds.  Weights come from numpy's legacy ``RandomState``
(bit-stable across numpy versions), so a fixture only has to store the seed.

Shapes mimic the real adapters:
  * ALBEF style (``ALBEF_attack/adv_attack.py:119-126``): ``[txt (layers*B, L, D), img (layers*B, N, D)]``
  * VLMO style (``vlmo/modules/vlmo_module.py:1387-1446``, batch 1):
    ``[pooled (1, D), cls_per_layer (layers, D), feats (layers, T+N, D)]``
  * MLM heads return ``[logits (B, L, 30522)]``.
"""
import numpy as np
import torch

VOCAB = 30522


def _w(seed, *shape, scale=1.0):
    a = np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * np.float32(scale)
    return torch.from_numpy(a)


class ToyWhiteBox:
    def __init__(self, device="cpu", hw=32, patch=8, dim=16, layers=3, text_len=6, seed=11):
        self.hw, self.patch, self.dim, self.layers, self.text_len = hw, patch, dim, layers, text_len
        self.ntok = (hw // patch) ** 2
        pin = 3 * patch * patch
        d = device
        self.w_patch = _w(seed + 0, pin, dim, scale=pin ** -0.5).to(d)
        self.pos = _w(seed + 1, 1, self.ntok + 1, dim, scale=0.5).to(d)
        self.cls = _w(seed + 2, 1, 1, dim, scale=0.5).to(d)
        self.w_img = [_w(seed + 10 + i, dim, dim, scale=dim ** -0.5).to(d) for i in range(layers)]
        self.u_img = [_w(seed + 20 + i, dim, dim, scale=dim ** -0.5).to(d) for i in range(layers)]
        self.w_txt = [_w(seed + 30 + i, dim, dim, scale=dim ** -0.5).to(d) for i in range(layers)]
        self.c_txt = [_w(seed + 40 + i, dim, dim, scale=dim ** -0.5).to(d) for i in range(layers)]
        self.word = _w(seed + 50, 64, dim, scale=1.0).to(d)          # toy vocabulary of 64 ids
        self.w_pool = _w(seed + 51, dim, dim, scale=dim ** -0.5).to(d)
        self.w_mlm = _w(seed + 52, dim, VOCAB, scale=dim ** -0.5).to(d)
        self.text_ids = torch.arange(text_len, device=d).unsqueeze(0) * 7 % 64  # (1, L)

    # ---- building blocks
    def embed_text(self, ids):
        return self.word[ids]

    def _tokens(self, x):
        b, p, g = x.shape[0], self.patch, self.hw // self.patch
        t = x.reshape(b, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(b, g * g, 3 * p * p)
        h = t @ self.w_patch
        h = torch.cat([self.cls.expand(b, -1, -1), h], dim=1) + self.pos
        return h

    def _run(self, x, text_embeds):
        h, t = self._tokens(x), text_embeds
        img, txt = [], []
        for i in range(self.layers):
            img.append(h)
            txt.append(t)
            ctx = h.mean(dim=1, keepdim=True)
            h = h + torch.tanh(h @ self.w_img[i] + ctx @ self.u_img[i])
            t = t + torch.tanh(t @ self.w_txt[i] + ctx @ self.c_txt[i])
        return img, txt, h, t

    def _text_for(self, x):
        return self.embed_text(self.text_ids.expand(x.shape[0], -1))

    # ---- ALBEF-style adapters
    def albef_feats(self, x):
        img, txt, _, _ = self._run(x, self._text_for(x))
        return [torch.cat(txt, dim=0), torch.cat(img, dim=0)]

    def albef_feats_vl(self, xs):
        img, txt, _, _ = self._run(xs[0], xs[1])
        return [torch.cat(txt, dim=0), torch.cat(img, dim=0)]

    def mlm_logits(self, x):
        _, _, _, t = self._run(x, self._text_for(x))
        return [t @ self.w_mlm]

    # ---- VLMO-style adapters (batch 1, layer axis plays the batch role in the loss)
    def _vlmo_pack(self, img, txt, h):
        feats = torch.stack([torch.cat([t, i], dim=1) for t, i in zip(txt, img)], dim=1)  # (B,layers,T+N,D)
        pooled = torch.tanh(h[:, 0] @ self.w_pool)
        cls_layers = torch.stack([i[:, 0] for i in img], dim=1)[0]                        # (layers, D)
        return pooled, cls_layers, feats[0]

    def vlmo_feats(self, x):
        img, txt, h, _ = self._run(x, self._text_for(x))
        return list(self._vlmo_pack(img, txt, h))

    def vlmo_feats_vl(self, xs):
        img, txt, h, _ = self._run(xs[0], xs[1])
        return list(self._vlmo_pack(img, txt, h))

    def vlmo_mixed(self, x):
        img, txt, h, t = self._run(x, self._text_for(x))
        _, cls_layers, feats = self._vlmo_pack(img, txt, h)
        return [t @ self.w_mlm, cls_layers, feats]


def toy_inputs(batch, hw=32, seed=3, eps=0.125):
    """Clean images in [-1, 1] and a start point inside the eps-ball (as numpy RandomState draws)."""
    r = np.random.RandomState(seed)
    x0 = r.uniform(-1, 1, (batch, 3, hw, hw)).astype(np.float32)
    eta = r.uniform(-eps, eps, x0.shape).astype(np.float32)
    return torch.from_numpy(x0), torch.from_numpy(eta)
