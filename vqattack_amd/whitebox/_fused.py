"""Pre-LN transformer encoder of the frozen white boxes WITHOUT an autograd graph inside: library GEMMs, the hand-written
attention (``csrc/attn.hip``) and the fused block glue of ``csrc/block.hip`` -- forward and input-gradient backward written
out stage by stage.

Reference computation: ``Block.forward`` of ``VLMO_VQAttack/vlmo/modules/multiway_transformer.py:184-201`` (shared
attention, modality-expert FFNs split at the text length, layer scale ``gamma_1`` / ``gamma_2``) and ALBEF's ViT block
(``ALBEF_attack/models/vit.py``; no layer scale, one FFN).  Weights are frozen, so the backward produces the INPUT gradient
only: two GEMMs per linear layer disappear against a training backward, and nothing but the layer inputs, the packed
``qkv``, the attention statistics and the pre-GELU activations is kept from the forward.

Why not autograd: eager execution runs every residual add, layer-scale multiply, LayerNorm, slice, ``cat`` and
gradient-accumulation add as its own full pass over the (B, S, D) residual stream (10.8 % of the attack's device time in
round 3).  Here a stage boundary is ONE kernel: ``vqa_ln_fwd`` = residual add of the previous branch + LayerNorm (+ the
text / image split the expert GEMMs need), ``vqa_ln_bwd`` = LayerNorm backward + the gradient of the residual path + the
loss kernel's gradient of that feature map (+ the layer-scaled branch gradient).  The whole encoder is one
``torch.autograd.Function`` whose outputs are the per-layer feature maps and the final normalised states, so callers
(the ``model_fn`` closures, the loss kernels' ``torch.autograd.backward(outputs, grads)``) see ordinary tensors.
"""
import torch

from .. import attention as _attn
from .. import ops


class LayerSpec:
    """Frozen parameters of one block, in the form the stages consume (plain fp32 device tensors)."""
    __slots__ = ("ln1", "wqkv", "bqkv", "wproj", "bproj", "gamma1", "ln2", "mlp", "gamma2", "eps")

    def __init__(self, ln1, wqkv, bqkv, wproj, bproj, gamma1, ln2, mlp, gamma2, eps):
        self.ln1, self.wqkv, self.bqkv, self.wproj, self.bproj = ln1, wqkv, bqkv, wproj, bproj
        self.gamma1, self.ln2, self.mlp, self.gamma2, self.eps = gamma1, ln2, mlp, gamma2, eps


class EncoderSpec:
    def __init__(self, layers, final_ln, final_eps, heads):
        self.layers, self.final_ln, self.final_eps, self.heads = layers, final_ln, final_eps, heads


def supported(dim, heads):
    """The hand-written stages cover the head size of every BASELINE configuration (64) and D <= 1024."""
    return dim % heads == 0 and dim // heads == _attn.HEAD_DIM and dim % 4 == 0 and dim <= 1024


def weights_key(module):
    """Fingerprint of the frozen weights a spec was built from: storage address and in-place version counter of every
    parameter.  Detected: ``load_state_dict`` (copying or ``assign=True``), ``.to()``, ``p.copy_()`` / any in-place op ON
    THE PARAMETER.  NOT detected: writes through ``p.data`` (``p.data.copy_(w)``, ``p.data.add_(1)``) -- ``.data`` is a
    detached alias with its own version counter, so the key stays equal and a cached spec that holds COPIES (ALBEF's
    packed q / k / v, contiguous copies of strided parameters) would keep the old values.  Code that updates weights
    that way calls ``invalidate_fused()`` on the model (``FrozenVlmo`` / ``FrozenAlbef``); the reference-checkpoint
    loaders do.  ~0.1 us per parameter, once per encoder pass."""
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


def _c(t):
    t = t.detach()
    return t if t.is_contiguous() else t.contiguous()


def _ln(mod):
    return (_c(mod.weight), _c(mod.bias))


def _mlp(mod):
    return (_c(mod.fc1.weight), _c(mod.fc1.bias), _c(mod.fc2.weight), _c(mod.fc2.bias))


def vlmo_spec(model):
    """``FrozenVlmo`` -> EncoderSpec.  Expert layers carry two LayerNorm / MLP sets (text, image), VL-FFN layers one."""
    layers = []
    for blk in model.blocks:
        if blk.mlp_vl is None:
            ln2, mlp = [_ln(blk.norm2_text), _ln(blk.norm2_imag)], [_mlp(blk.mlp_text), _mlp(blk.mlp_imag)]
        else:
            ln2, mlp = [_ln(blk.norm2_vl)], [_mlp(blk.mlp_vl)]
        layers.append(LayerSpec(_ln(blk.norm1), _c(blk.attn.qkv.weight), _c(blk.attn.qkv.bias), _c(blk.attn.proj.weight),
                                _c(blk.attn.proj.bias), _c(blk.gamma_1), ln2, mlp, _c(blk.gamma_2), blk.norm1.eps))
    return EncoderSpec(layers, _ln(model.norm), model.norm.eps, model.cfg.heads)


def vit_spec(blocks, final_norm, heads):
    """ALBEF's ViT blocks (separate q / k / v projections packed into one GEMM once -- the weights are frozen)."""
    layers = []
    for blk in blocks:
        a = blk.attn
        wqkv = torch.cat([a.q.weight.detach(), a.k.weight.detach(), a.v.weight.detach()], dim=0).contiguous()
        bqkv = torch.cat([a.q.bias.detach(), a.k.bias.detach(), a.v.bias.detach()], dim=0).contiguous()
        layers.append(LayerSpec(_ln(blk.norm1), wqkv, bqkv, _c(a.o.weight), _c(a.o.bias), None, [_ln(blk.norm2)],
                                [_mlp(blk.mlp)], None, blk.norm1.eps))
    return EncoderSpec(layers, _ln(final_norm), final_norm.eps, heads)


class _Call:
    """Per-call context: the attention masks of the text batch and the token layout."""

    def __init__(self, spec, biases, n_text):
        self.spec, self.biases, self.n_text = spec, biases, n_text


def _attention_inputs(qkv5, bias, need_grad):
    q, k, v = qkv5[:, :, 0], qkv5[:, :, 1], qkv5[:, :, 2]
    b, s, h, _ = q.shape
    hole = None
    if isinstance(bias, _attn.KeyHoleBias):
        if need_grad and not _attn.scores_fit(b, h, s, s):
            bias = bias.dense()
        else:
            bias, hole = bias.slab, bias.hole
    bias_t, bstr = _attn._bias_view(None if bias is None else bias.detach(), b, h, s, s)
    return q, k, v, bias_t, bstr, hole


def _forward(x0, call, save):
    spec = call.spec
    b, s, d = x0.shape
    rows, t = b * s, call.n_text
    dev, f32 = x0.device, torch.float32
    scale = _attn.HEAD_DIM ** -0.5
    x = x0 if x0.is_contiguous() else x0.contiguous()
    pend = None                       # (m0, m1, gamma2) of the previous layer: its residual add happens in the next LN
    feats, saved = [], []
    for li, lay in enumerate(spec.layers):
        y = torch.empty(rows, d, dtype=f32, device=dev)
        mean1, rstd1 = torch.empty(rows, dtype=f32, device=dev), torch.empty(rows, dtype=f32, device=dev)
        if pend is None:
            x_l = x
            ops.ln_fwd(x, lay.ln1[0], lay.ln1[1], y, mean1, rstd1, lay.eps)
        else:
            x_l = torch.empty(b, s, d, dtype=f32, device=dev)
            ops.ln_fwd(x, lay.ln1[0], lay.ln1[1], y, mean1, rstd1, lay.eps, r0=pend[0], r1=pend[1], rscale=pend[2],
                       x_out=x_l, period=s if pend[1] is not None else 0, split=t if pend[1] is not None else 0)
            feats.append(x_l)
        qkv = torch.addmm(lay.bqkv, y, lay.wqkv.t())
        del y
        qkv5 = qkv.view(b, s, 3, spec.heads, _attn.HEAD_DIM)
        q, k, v, bias_t, bstr, hole = _attention_inputs(qkv5, None if call.biases is None else call.biases[li], save)
        o, lse, scores = _attn._forward(q, k, v, bias_t, bstr, scale, save_scores=save, key_hole=hole)
        p = torch.addmm(lay.bproj, o.view(rows, d), lay.wproj.t())
        x1 = torch.empty(b, s, d, dtype=f32, device=dev)
        mean2, rstd2 = torch.empty(rows, dtype=f32, device=dev), torch.empty(rows, dtype=f32, device=dev)
        two = len(lay.mlp) == 2
        if two:
            ys = [torch.empty(b * t, d, dtype=f32, device=dev), torch.empty(b * (s - t), d, dtype=f32, device=dev)]
            ops.ln_fwd(x_l, lay.ln2[0][0], lay.ln2[0][1], ys[0], mean2, rstd2, lay.eps, r0=p, rscale=lay.gamma1, x_out=x1,
                       gamma1=lay.ln2[1][0], beta1=lay.ln2[1][1], y1=ys[1], period=s, split=t)
        else:
            ys = [torch.empty(rows, d, dtype=f32, device=dev)]
            ops.ln_fwd(x_l, lay.ln2[0][0], lay.ln2[0][1], ys[0], mean2, rstd2, lay.eps, r0=p, rscale=lay.gamma1, x_out=x1)
        del p
        hs, ms = [], []
        for ye, (w1, b1, w2, b2) in zip(ys, lay.mlp):
            h = torch.addmm(b1, ye, w1.t())
            a = ops.gelu_fwd(h)
            ms.append(torch.addmm(b2, a, w2.t()))
            hs.append(h)
            del a
        del ys
        if save:
            saved.append(dict(x_l=x_l, mean1=mean1, rstd1=rstd1, qkv=qkv, o=o, lse=lse, scores=scores, bias=bias_t,
                              bstr=bstr, x1=x1, mean2=mean2, rstd2=rstd2, hs=hs))
        x, pend = x1, (ms[0], ms[1] if two else None, lay.gamma2)
    x_last = torch.empty(b, s, d, dtype=f32, device=dev)
    states = torch.empty(b, s, d, dtype=f32, device=dev)
    mean_f, rstd_f = torch.empty(rows, dtype=f32, device=dev), torch.empty(rows, dtype=f32, device=dev)
    ops.ln_fwd(x, spec.final_ln[0], spec.final_ln[1], states, mean_f, rstd_f, spec.final_eps, r0=pend[0], r1=pend[1],
               rscale=pend[2], x_out=x_last, period=s if pend[1] is not None else 0, split=t if pend[1] is not None else 0)
    feats.append(x_last)
    if save:
        saved.append(dict(x_l=x_last, mean1=mean_f, rstd1=rstd_f))
    return feats, states, saved


def _dense(g, like):
    """A gradient autograd hands over, as the kernels read it: fp32, contiguous, or None."""
    if g is None:
        return None
    return g if g.is_contiguous() else g.contiguous()


def _branch_buffers(two, b, s, t, d, dev):
    if two:
        return [torch.empty(b * t, d, dtype=torch.float32, device=dev),
                torch.empty(b * (s - t), d, dtype=torch.float32, device=dev)]
    return [torch.empty(b * s, d, dtype=torch.float32, device=dev)]


def _backward(saved, call, g_feats, g_states, shape):
    """``g_feats[l]``: gradient of feature map l + 1 (the output of block l; None = no loss on it); ``g_states``: gradient
    of the final normalised states.  Returns the gradient of the encoder's input."""
    spec = call.spec
    b, s, d = shape
    rows, t = b * s, call.n_text
    dev = saved[-1]["x_l"].device
    scale = _attn.HEAD_DIM ** -0.5
    n = len(spec.layers)
    fin = saved[n]
    if g_states is None:
        g_states = torch.zeros(b, s, d, dtype=torch.float32, device=dev)
    last = spec.layers[n - 1]
    two = len(last.mlp) == 2
    dx = torch.empty(rows, d, dtype=torch.float32, device=dev)
    dm = _branch_buffers(two, b, s, t, d, dev) if (two or last.gamma2 is not None) else [dx]
    ops.ln_bwd(_dense(g_states, dx), fin["x_l"], fin["mean1"], fin["rstd1"], spec.final_ln[0], dx,
               g_inj=_dense(g_feats[n - 1], dx), rscale=last.gamma2, dr0=dm[0] if dm[0] is not dx else None,
               dr1=dm[1] if two else None, period=s if two else 0, split=t if two else 0)
    g = dx                                        # gradient w.r.t. the last layer's x1 (residual path)
    for li in range(n - 1, -1, -1):
        lay, sv = spec.layers[li], saved[li]
        two = len(lay.mlp) == 2
        dys = []
        for dme, h, (w1, _b1, w2, _b2) in zip(dm, sv["hs"], lay.mlp):
            da = torch.mm(dme, w2)                # (rows_e, 4D)
            ops.gelu_bwd(h, da)                   # in place: dh
            dys.append(torch.mm(da, w1))
            del da
        del dm
        dx1 = torch.empty(rows, d, dtype=torch.float32, device=dev)
        dp = torch.empty(rows, d, dtype=torch.float32, device=dev) if lay.gamma1 is not None else dx1
        ops.ln_bwd(dys[0], sv["x1"], sv["mean2"], sv["rstd2"], lay.ln2[0][0], dx1, dy1=dys[1] if two else None,
                   gamma1=lay.ln2[1][0] if two else None, g_a=g, rscale=lay.gamma1,
                   dr0=dp if lay.gamma1 is not None else None, period=s if two else 0, split=t if two else 0)
        del dys, g
        do = torch.mm(dp, lay.wproj)
        dqkv = torch.empty_like(sv["qkv"])
        qkv5, dqkv5 = sv["qkv"].view(b, s, 3, spec.heads, _attn.HEAD_DIM), dqkv.view(b, s, 3, spec.heads, _attn.HEAD_DIM)
        _attn._backward(qkv5[:, :, 0], qkv5[:, :, 1], qkv5[:, :, 2], sv["bias"], sv["bstr"], sv["o"], sv["lse"],
                        do.view(b, s, spec.heads, _attn.HEAD_DIM), dqkv5[:, :, 0], dqkv5[:, :, 1], dqkv5[:, :, 2], scale,
                        scores=sv["scores"])
        del do, dp
        dy1 = torch.mm(dqkv, lay.wqkv)
        del dqkv
        dx_l = torch.empty(rows, d, dtype=torch.float32, device=dev)
        if li > 0:
            prev = spec.layers[li - 1]
            ptwo = len(prev.mlp) == 2
            dm = _branch_buffers(ptwo, b, s, t, d, dev) if (ptwo or prev.gamma2 is not None) else [dx_l]
            ops.ln_bwd(dy1, sv["x_l"], sv["mean1"], sv["rstd1"], lay.ln1[0], dx_l, g_a=dx1,
                       g_inj=_dense(g_feats[li - 1], dx_l), rscale=prev.gamma2, dr0=dm[0] if dm[0] is not dx_l else None,
                       dr1=dm[1] if ptwo else None, period=s if ptwo else 0, split=t if ptwo else 0)
        else:
            ops.ln_bwd(dy1, sv["x_l"], sv["mean1"], sv["rstd1"], lay.ln1[0], dx_l, g_a=dx1)
        del dy1, dx1
        saved[li] = None                          # this layer's activations are dead
        g = dx_l
    return g.view(b, s, d)


class _Encoder(torch.autograd.Function):
    """(x0 (B, S, D), call) -> (feature map 1, ..., feature map L, final normalised states)."""

    @staticmethod
    def forward(ctx, x0, call):
        # an output the loss does not touch (every feature map during an MLM step of the dual loss) must reach backward
        # as None, not as a materialised zero tensor: 12 x (fill + read) of (B, S, D) per step otherwise
        ctx.set_materialize_grads(False)
        save = ctx.needs_input_grad[0]
        feats, states, saved = _forward(x0.detach(), call, save)
        if save:
            # the layer inputs x_l are the Function's own input / outputs: they must go through save_for_backward (an
            # output kept in a ctx attribute is a reference cycle through the C++ graph that no collector sees)
            layer_inputs = [sv.pop("x_l") for sv in saved]
            ctx.save_for_backward(*layer_inputs)
        ctx.call, ctx.saved, ctx.shape = call, saved, tuple(x0.shape)
        return tuple(feats) + (states,)

    @staticmethod
    def backward(ctx, *grads):
        saved, ctx.saved = ctx.saved, None
        if saved is None:
            raise RuntimeError("the fused encoder's backward ran twice (its activations are freed by the first run)")
        for sv, x_l in zip(saved, ctx.saved_tensors):
            sv["x_l"] = x_l
        return _backward(saved, ctx.call, grads[:-1], grads[-1], ctx.shape), None


def encode(x0, spec, biases, n_text):
    """Run the encoder on the residual stream ``x0`` (B, S, D) (token embeddings + type embeddings).  ``biases``: one
    additive attention mask per layer (or None), ``n_text``: text tokens at the front of the sequence (0 = one
    modality).  Returns ``([x0, feature maps 1..L], states)`` like the eager block loop."""
    if not x0.is_cuda or x0.dtype != torch.float32:
        raise ops._hip.HipExtensionError("the fused encoder runs on fp32 HIP tensors")
    outs = _Encoder.apply(x0, _Call(spec, biases, n_text))
    return [x0] + list(outs[:-1]), outs[-1]
