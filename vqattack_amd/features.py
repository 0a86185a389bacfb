"""Per-layer feature lists that are consumed WITHOUT being concatenated.

The reference's adapters pack the 13 (25 for VLMO-large) per-layer activations with ``torch.cat(..., axis=0)``
(``ALBEF_attack/adv_attack.py:124-125``) or ``torch.stack(..., axis=1)[0]`` (``vlmo_module.py:1435-1444``) before the
loss sees them -- at batch 256 that is 2 x 5.9 GB of pure copy per forward and the same again in the backward
(SURVEY.md section 8f, rank 1).  ``LayerFeatures`` stands for the same logical tensor, ``cat(layers, dim=0)`` of shape
``(L*B, N, D)``, but keeps the L tensors as they come out of the encoder; ``attacks._feature_loss_backward`` launches
the fused loss kernel once per layer and hands each gradient straight to autograd, so no packed copy is ever made.

``row_weight`` (uint8 ``(B, N)``, optional) replaces the VLMO adapter's ragged drop of padded text tokens
(``feats_list_text = target_feats[0, :, text_masks[0]]``): weight 0 = padded token (skipped), 1 = token, 2 = the [CLS]
row that the VLMO loss counts both alone and as a token.
"""
import torch


class LayerFeatures:
    def __init__(self, layers, row_weight=None):
        layers = list(layers)
        if not layers:
            raise ValueError("LayerFeatures needs at least one layer")
        shape = layers[0].shape
        if len(shape) != 3 or any(t.shape != shape for t in layers):
            raise ValueError("all layers must share one (B, N, D) shape")
        if row_weight is not None:
            if row_weight.dtype != torch.uint8 or tuple(row_weight.shape) != tuple(shape[:2]):
                raise ValueError("row_weight must be uint8 of shape (B, N) = {}".format(tuple(shape[:2])))
        self.layers = layers
        self.row_weight = row_weight

    # --- the little bit of tensor interface the attack operators use on model outputs / targets
    @property
    def shape(self):
        b, n, d = self.layers[0].shape
        return torch.Size((len(self.layers) * b, n, d))

    @property
    def device(self):
        return self.layers[0].device

    @property
    def is_cuda(self):
        return self.layers[0].is_cuda

    def __len__(self):
        return len(self.layers)

    def __getitem__(self, index):
        """Only the reference's token truncation ``[:, :n, :]`` is meaningful on a layer list."""
        if (isinstance(index, tuple) and len(index) == 3 and index[0] == slice(None) and index[2] == slice(None)
                and isinstance(index[1], slice) and index[1].start in (None, 0) and index[1].step in (None, 1)):
            n = index[1].stop
            w = None if self.row_weight is None else self.row_weight[:, :n].contiguous()
            return LayerFeatures([t[:, :n, :] for t in self.layers], w)
        raise IndexError("LayerFeatures supports only [:, :n, :]")

    def rows(self, n):
        """The first ``n`` samples of every layer (views) -- a batch prefix, for schedules where samples finish early."""
        w = None if self.row_weight is None else self.row_weight[:n].contiguous()
        return LayerFeatures([t[:n] for t in self.layers], w)

    def detach(self):
        return LayerFeatures([t.detach() for t in self.layers], self.row_weight)

    def materialize(self):
        """The packed tensor the reference would have built (for tests / interop; costs the copy)."""
        return torch.cat(self.layers, dim=0)


def layer_pairs(out, target):
    """Zip a model output with its target into per-launch ``(out_l, target_l, row_weight)`` triples.

    Either side may be a ``LayerFeatures`` or a packed tensor ``(L*B, N, D)``; a packed side is viewed per layer
    (no copy).  Two plain tensors come back as a single triple.
    """
    o_lf, t_lf = isinstance(out, LayerFeatures), isinstance(target, LayerFeatures)
    if not o_lf and not t_lf:
        return [(out, target, None)]
    ref = out if o_lf else target
    n_layers, b = len(ref), ref.layers[0].shape[0]

    def per_layer(v, is_lf):
        if is_lf:
            if len(v) != n_layers:
                raise ValueError("layer count mismatch: {} vs {}".format(len(v), n_layers))
            return v.layers
        if v.shape[0] != n_layers * b:
            raise ValueError("packed tensor has {} rows, expected L*B = {}".format(v.shape[0], n_layers * b))
        return [v[i * b:(i + 1) * b] for i in range(n_layers)]

    weight = out.row_weight if o_lf and out.row_weight is not None else (target.row_weight if t_lf else None)
    return [(o, t, weight) for o, t in zip(per_layer(out, o_lf), per_layer(target, t_lf))]
