"""Tensor-level wrappers over the C ABI (one function per kernel entry point).

Every function launches on the current HIP stream of its tensors' device and returns without synchronising.
``out=`` arguments let the PGD loop ping-pong between two pre-allocated image buffers.
"""
import contextlib
import ctypes

import torch

from . import _hip
from ._hip import VQA_CHECK_RANGE, VQA_CLIP, check, dev_f32, lib, ptr, same_device, stream_for

_INF = float("inf")


@contextlib.contextmanager
def _on(t):
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        with torch.cuda.device(t.device):
            yield
    else:
        yield


def _clip_args(clip_min, clip_max):
    """(mode bits, cmin, cmax); a missing bound becomes +-inf (torch.clamp with one bound)."""
    if clip_min is None and clip_max is None:
        return 0, -_INF, _INF
    return VQA_CLIP, (-_INF if clip_min is None else float(clip_min)), (_INF if clip_max is None else float(clip_max))


def _out_like(x, out):
    if out is None:
        return torch.empty_like(x, memory_format=torch.contiguous_format)
    dev_f32(out, "out")
    if out.shape != x.shape:
        raise ValueError("out has shape {}, expected {}".format(tuple(out.shape), tuple(x.shape)))
    return out


def new_flag(device):
    """int32[1] device word the kernels OR range violations into (read once at the end of an attack)."""
    return torch.zeros(1, dtype=torch.int32, device=device)


# ----------------------------------------------------------------------------------------- L-inf
def linf_init(x, eta, eps, clip_min, clip_max, flag=None, out=None):
    dev_f32(x, "x")
    if eta is not None:
        dev_f32(eta, "eta")
        if eta.shape != x.shape:
            raise ValueError("eta shape {} != x shape {}".format(tuple(eta.shape), tuple(x.shape)))
    same_device(x, eta, out, flag)
    out = _out_like(x, out)
    mode, lo, hi = _clip_args(clip_min, clip_max)
    if flag is not None and mode:
        mode |= VQA_CHECK_RANGE
    if x.numel() == 0:
        return out
    with _on(x):
        check(lib().vqa_linf_init(ptr(x), ptr(eta), ptr(out), x.numel(), eps, lo, hi, mode, ptr(flag),
                                  stream_for(x)), "vqa_linf_init")
    return out


def linf_fgm(x, g, eps_iter, clip_min, clip_max, flag=None, out=None):
    dev_f32(x, "x"), dev_f32(g, "grad")
    if g.shape != x.shape:
        raise ValueError("grad shape {} != x shape {}".format(tuple(g.shape), tuple(x.shape)))
    same_device(x, g, out, flag)
    out = _out_like(x, out)
    mode, lo, hi = _clip_args(clip_min, clip_max)
    if flag is not None and mode:
        mode |= VQA_CHECK_RANGE
    if x.numel() == 0:
        return out
    with _on(x):
        check(lib().vqa_linf_fgm(ptr(x), ptr(g), ptr(out), x.numel(), eps_iter, lo, hi, mode, ptr(flag),
                                 stream_for(x)), "vqa_linf_fgm")
    return out


def linf_step(x, g, x0, eps_iter, eps, clip_min, clip_max, flag=None, out=None):
    """The fused north-star kernel: FGM update + eps-ball projection + clamp, 16 B/element."""
    dev_f32(x, "x"), dev_f32(g, "grad"), dev_f32(x0, "ori_x")
    if g.shape != x.shape or x0.shape != x.shape:
        raise ValueError("shape mismatch: x {}, grad {}, ori_x {}".format(tuple(x.shape), tuple(g.shape),
                                                                          tuple(x0.shape)))
    same_device(x, g, x0, out, flag)
    out = _out_like(x, out)
    mode, lo, hi = _clip_args(clip_min, clip_max)
    if flag is not None and mode:
        mode |= VQA_CHECK_RANGE
    if x.numel() == 0:
        return out
    with _on(x):
        check(lib().vqa_linf_step(ptr(x), ptr(g), ptr(x0), ptr(out), x.numel(), eps_iter, eps, lo, hi, mode,
                                  ptr(flag), stream_for(x)), "vqa_linf_step")
    return out


def linf_project(adv, x0, eps, clip_min, clip_max, out=None):
    dev_f32(adv, "adv_x"), dev_f32(x0, "ori_x")
    if adv.shape != x0.shape:
        raise ValueError("shape mismatch: adv {}, ori_x {}".format(tuple(adv.shape), tuple(x0.shape)))
    same_device(adv, x0, out)
    out = _out_like(adv, out)
    mode, lo, hi = _clip_args(clip_min, clip_max)
    if adv.numel() == 0:
        return out
    with _on(adv):
        check(lib().vqa_linf_project(ptr(adv), ptr(x0), ptr(out), adv.numel(), eps, lo, hi, mode,
                                     stream_for(adv)), "vqa_linf_project")
    return out


def clip_eta_linf(eta, eps):
    dev_f32(eta, "eta")
    out = torch.empty_like(eta)
    if eta.numel() == 0:
        return out
    with _on(eta):
        check(lib().vqa_clip_eta_linf(ptr(eta), ptr(out), eta.numel(), eps, stream_for(eta)), "vqa_clip_eta_linf")
    return out


def optimize_linear_linf(grad, eps):
    dev_f32(grad, "grad")
    out = torch.empty_like(grad)
    if grad.numel() == 0:
        return out
    with _on(grad):
        check(lib().vqa_optimize_linear_linf(ptr(grad), ptr(out), grad.numel(), eps, stream_for(grad)),
              "vqa_optimize_linear_linf")
    return out


def zero_out_clipped_grads(grad, x, clip_min, clip_max):
    dev_f32(grad, "grad"), dev_f32(x, "x")
    if grad.shape != x.shape:
        raise ValueError("grad shape {} != x shape {}".format(tuple(grad.shape), tuple(x.shape)))
    out = torch.empty_like(grad)
    if grad.numel() == 0:
        return out
    with _on(grad):
        check(lib().vqa_zero_out_clipped_grads(ptr(grad), ptr(x), ptr(out), grad.numel(), float(clip_min),
                                               float(clip_max), stream_for(grad)), "vqa_zero_out_clipped_grads")
    return out


# ----------------------------------------------------------------------------------------- per-sample norms
def _per_sample(t):
    batch = t.shape[0] if t.dim() > 0 else 1
    n_per = t.numel() // max(batch, 1)
    return batch, n_per


def _workspace(t):
    batch, n_per = _per_sample(t)
    nbytes = lib().vqa_reduce_ws_bytes(batch, n_per)
    return torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=t.device)


def sumsq_per_sample(t, sub=None):
    """out[b] = sum((t[b] - sub[b])**2), deterministic two-stage reduction."""
    dev_f32(t, "t")
    if sub is not None:
        dev_f32(sub, "sub")
        if sub.shape != t.shape:
            raise ValueError("shape mismatch")
    batch, n_per = _per_sample(t)
    out = torch.empty(batch, dtype=torch.float32, device=t.device)
    if t.numel() == 0:
        return out.zero_()
    ws = _workspace(t)
    with _on(t):
        check(lib().vqa_sumsq_per_sample(ptr(t), ptr(sub), ptr(out), batch, n_per, ptr(ws), stream_for(t)),
              "vqa_sumsq_per_sample")
    return out


def absmax_ties_per_sample(g):
    dev_f32(g, "grad")
    batch, n_per = _per_sample(g)
    amax = torch.empty(batch, dtype=torch.float32, device=g.device)
    ties = torch.empty(batch, dtype=torch.float32, device=g.device)
    if g.numel() == 0:
        return amax.zero_(), ties.zero_()
    ws = _workspace(g)
    with _on(g):
        check(lib().vqa_absmax_ties_per_sample(ptr(g), ptr(amax), ptr(ties), batch, n_per, ptr(ws), stream_for(g)),
              "vqa_absmax_ties_per_sample")
    return amax, ties


def l2_fgm(x, g, eps_iter, clip_min, clip_max, flag=None, out=None, check_range=True):
    """``flag``: receives VQA_FLAG_DEGENERATE when a sample's gradient norm is not finite (the reference's self-check in
    optimize_linear, utils.py:110-116) and, with ``check_range``, VQA_FLAG_RANGE for inputs outside the clip range."""
    dev_f32(x, "x"), dev_f32(g, "grad")
    if g.shape != x.shape:
        raise ValueError("shape mismatch")
    ss = sumsq_per_sample(g)
    out = _out_like(x, out)
    batch, n_per = _per_sample(x)
    mode, lo, hi = _clip_args(clip_min, clip_max)
    if flag is not None and mode and check_range:
        mode |= VQA_CHECK_RANGE
    if x.numel() == 0:
        return out
    with _on(x):
        check(lib().vqa_l2_fgm(ptr(x), ptr(g), ptr(ss), ptr(out), batch, n_per, eps_iter, lo, hi, mode, ptr(flag),
                               stream_for(x)), "vqa_l2_fgm")
    return out


def l2_project(adv, x0, eps, clip_min, clip_max, out=None):
    dev_f32(adv, "adv_x"), dev_f32(x0, "ori_x")
    if adv.shape != x0.shape:
        raise ValueError("shape mismatch")
    ss = sumsq_per_sample(adv, sub=x0)
    out = _out_like(adv, out)
    batch, n_per = _per_sample(adv)
    mode, lo, hi = _clip_args(clip_min, clip_max)
    if adv.numel() == 0:
        return out
    with _on(adv):
        check(lib().vqa_l2_project(ptr(adv), ptr(x0), ptr(ss), ptr(out), batch, n_per, eps, lo, hi, mode,
                                   stream_for(adv)), "vqa_l2_project")
    return out


def l1_fgm(x, g, eps_iter, clip_min, clip_max, flag=None, out=None, check_range=True):
    """``flag``: receives VQA_FLAG_DEGENERATE when a sample's largest |gradient| is 0 or NaN (utils.py:101-104)."""
    dev_f32(x, "x"), dev_f32(g, "grad")
    if g.shape != x.shape:
        raise ValueError("shape mismatch")
    amax, ties = absmax_ties_per_sample(g)
    out = _out_like(x, out)
    batch, n_per = _per_sample(x)
    mode, lo, hi = _clip_args(clip_min, clip_max)
    if flag is not None and mode and check_range:
        mode |= VQA_CHECK_RANGE
    if x.numel() == 0:
        return out
    with _on(x):
        check(lib().vqa_l1_fgm(ptr(x), ptr(g), ptr(amax), ptr(ties), ptr(out), batch, n_per, eps_iter, lo, hi, mode,
                               ptr(flag), stream_for(x)), "vqa_l1_fgm")
    return out


def scale_per_sample(t, stat, stat2, eps, kind, out=None, flag=None):
    """``flag`` (int32 device word): kinds 1 / 2 OR ``VQA_FLAG_DEGENERATE`` into it where the reference's unit-norm
    self-check of ``optimize_linear`` would fail (utils.py:101-104, :110-116)."""
    dev_f32(t, "t"), dev_f32(stat, "stat")
    out = _out_like(t, out)
    batch, n_per = _per_sample(t)
    if t.numel() == 0:
        return out
    with _on(t):
        check(lib().vqa_scale_per_sample(ptr(t), ptr(stat), ptr(stat2), ptr(out), batch, n_per, eps, kind, ptr(flag),
                                         stream_for(t)), "vqa_scale_per_sample")
    return out


# ----------------------------------------------------------------------------------------- loss
_COS_EPS = 1e-6   # nn.CosineSimilarity(eps=1e-6) in the reference
_partials = {}


def _partial_buf(device):
    """Per-(device, stream) scratch for the loss partials: launches on one stream are ordered, so reuse is safe."""
    # zero-initialised: the last word is the arrival counter of the in-kernel fold (the folding workgroup resets it)
    if torch.cuda.is_current_stream_capturing():
        # inside hipGraph capture the scratch must live in that graph's private pool: never cache it
        return torch.zeros(lib().vqa_neg_cos_partials(), dtype=torch.float32, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _partials.get(key)
    if buf is None:
        buf = torch.zeros(lib().vqa_neg_cos_partials(), dtype=torch.float32, device=device)
        _partials[key] = buf
    return buf


class Workspace:
    """Scratch tensors one attack call owns and reuses across its iterations: the loss kernels' gradient buffers
    (4.5 GB per launch at VLMO-base batch 64) and the cross-entropy row buffers.  A buffer handed out here is dead once
    ``torch.autograd.backward`` has consumed it, i.e. before the next iteration asks again for the same key."""

    def __init__(self):
        self._bufs = {}

    def get(self, key, shape, dtype, device, zero=False):
        """A buffer per ROLE (``key``), not per shape: one flat allocation sized for the largest request so far, handed
        out as a prefix view -- a driver whose active batch shrinks (``attack_mixed``: finished samples leave the batch)
        keeps ONE set of buffers instead of one per batch size.  ``zero=True``: zero-filled when (re)allocated, and only
        then; a smaller request of the same role sees the same leading rows."""
        if torch.cuda.is_current_stream_capturing():
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
        numel = 1
        for n in shape:
            numel *= int(n)
        full = (key, dtype, device)
        buf = self._bufs.get(full)
        if buf is None or buf.numel() < numel:
            buf = (torch.zeros if zero else torch.empty)(max(numel, 1), dtype=dtype, device=device)
            self._bufs[full] = buf
        return buf[:numel].view(shape)

    def persistent(self):
        """True when a buffer handed out under a key is the SAME memory on the next request (not during graph capture)."""
        return not torch.cuda.is_current_stream_capturing()


def _scratch(ws, key, shape, dtype, device):
    return torch.empty(shape, dtype=dtype, device=device) if ws is None else ws.get(key, shape, dtype, device)


def _rows_view(t, name):
    """(rows0, rows1, D, stride0, stride1) of a 2-d or 3-d fp32 tensor whose last dim is dense."""
    if t.dim() == 2:
        r0, r1, d = 1, t.shape[0], t.shape[1]
        s0, s1 = 0, t.stride(0)
    elif t.dim() == 3:
        r0, r1, d = t.shape
        s0, s1 = t.stride(0), t.stride(1)
    else:
        raise ValueError("{} must be 2-d (rows, D) or 3-d (rows0, rows1, D), got {}".format(name, tuple(t.shape)))
    return r0, r1, d, s0, s1


def _kernel_ready(t):
    if t.dim() < 2:
        return False
    if t.stride(-1) != 1 and t.shape[-1] > 1:
        return False
    if t.data_ptr() % 16:
        return False
    return all(s % 4 == 0 for s in t.stride()[:-1])


def neg_cos_rows(a, b, loss_out, accumulate, gscale=1.0, want_grad=True, row_weight=None, weight_period=1, ws=None):
    """loss_out[0] (+)= gscale * sum_rows -cos(a_row, b_row); returns d(that)/d a (same shape as ``a``) or None.

    ``a`` / ``b`` may be strided views (the reference's ``[:, :feat_len, :]`` truncations) as long as rows are
    dense.  ``loss_out`` is a 1-element fp32 device tensor (a slot of the attack's loss buffer).
    ``row_weight`` (uint8, ``weight_period * rows1`` entries): per-row integer weight, indexed by
    ``(outer % weight_period, inner)``; 0 drops a row (padded token), 2 counts it twice.
    """
    dev_f32(a, "out", contiguous=False), dev_f32(b, "y", contiguous=False)
    if a.shape != b.shape:
        raise ValueError("feature/target shape mismatch: {} vs {}".format(tuple(a.shape), tuple(b.shape)))
    d = a.shape[-1]
    if d % 4 or d > 2048:
        raise _hip.HipExtensionError("feature dim {} unsupported by vqa_neg_cos_rows (multiple of 4, <= 2048)".format(d))
    if not _kernel_ready(a):
        a = a.contiguous()
    if not _kernel_ready(b):
        b = b.contiguous()
    r0, r1, d, a0, a1 = _rows_view(a, "out")
    _, _, _, b0, b1 = _rows_view(b, "y")
    ga = None
    g0 = g1 = 0
    if want_grad:
        ga = _scratch(ws, "cos1", a.shape, torch.float32, a.device)
        _, _, _, g0, g1 = _rows_view(ga, "grad")
    if row_weight is not None:
        if row_weight.dtype != torch.uint8 or not row_weight.is_cuda or not row_weight.is_contiguous():
            raise TypeError("row_weight must be a contiguous uint8 device tensor")
        if row_weight.numel() != weight_period * r1:
            raise ValueError("row_weight has {} entries, expected weight_period*rows1 = {}".format(
                row_weight.numel(), weight_period * r1))
    if a.numel() == 0:                 # no rows: zero loss contribution, empty gradient
        if not accumulate and loss_out is not None:
            loss_out.zero_()
        return ga
    part = _partial_buf(a.device)
    with _on(a):
        st = stream_for(a)
        check(lib().vqa_neg_cos_rows(ptr(a), ptr(b), ptr(ga), ptr(part), ptr(row_weight), weight_period, r0, r1, d,
                                     a0, a1, b0, b1, g0, g1, gscale, _COS_EPS, ptr(loss_out), 1 if accumulate else 0,
                                     st), "vqa_neg_cos_rows")      # loss folded by the last workgroup: one launch
    return ga


class SubWorkspace:
    """A view of a Workspace whose keys are prefixed (per-layer fallback launches must not share one buffer)."""

    def __init__(self, ws, tag):
        self.ws, self.tag = ws, tag

    def get(self, key, shape, dtype, device, zero=False):
        return self.ws.get((self.tag, key), shape, dtype, device, zero=zero)

    def persistent(self):
        return self.ws.persistent()


def neg_cos_rows_multi(a_list, b_list, loss_out, accumulate, gscale=1.0, want_grad=True, row_weight=None,
                       weight_period=1, ws=None):
    """``neg_cos_rows`` over a list of same-shaped feature maps in ONE launch; returns the list of gradients (views of
    one ``(L, ...)`` buffer) or None.  Falls back to per-map launches when the maps do not share shape and strides."""
    n = len(a_list)
    if n != len(b_list) or n == 0:
        raise ValueError("need as many targets as feature maps")
    a0, b0 = a_list[0], b_list[0]
    uniform = (n <= lib().vqa_neg_cos_max_layers() and a0.numel() > 0 and
               all(t.shape == a0.shape and t.stride() == a0.stride() and t.dtype == torch.float32 and t.is_cuda and
                   _kernel_ready(t) for t in a_list) and
               all(t.shape == a0.shape and t.stride() == b0.stride() and t.dtype == torch.float32 and t.is_cuda and
                   _kernel_ready(t) for t in b_list) and a0.shape[-1] % 4 == 0 and a0.shape[-1] <= 2048)
    if not uniform or n == 1:
        grads = []
        for k, (a, b) in enumerate(zip(a_list, b_list)):
            grads.append(neg_cos_rows(a, b, loss_out, accumulate or k > 0, gscale, want_grad, row_weight, weight_period,
                                      ws=None if ws is None else SubWorkspace(ws, k)))
        return grads if want_grad else None
    r0, r1, d, sa0, sa1 = _rows_view(a0, "out")
    _, _, _, sb0, sb1 = _rows_view(b0, "y")
    ga = None
    g0 = g1 = 0
    if want_grad:
        ga = _scratch(ws, "cosN", (n,) + tuple(a0.shape), torch.float32, a0.device)
        _, _, _, g0, g1 = _rows_view(ga[0], "grad")
    if row_weight is not None:
        if row_weight.dtype != torch.uint8 or not row_weight.is_cuda or not row_weight.is_contiguous():
            raise TypeError("row_weight must be a contiguous uint8 device tensor")
        if row_weight.numel() != weight_period * r1:
            raise ValueError("row_weight has {} entries, expected weight_period*rows1 = {}".format(
                row_weight.numel(), weight_period * r1))
    arr = ctypes.c_void_p * n
    pa = arr(*[t.data_ptr() for t in a_list])
    pb = arr(*[t.data_ptr() for t in b_list])
    pg = arr(*[ga[i].data_ptr() for i in range(n)]) if want_grad else None
    part = _partial_buf(a0.device)
    with _on(a0):
        st = stream_for(a0)
        check(lib().vqa_neg_cos_rows_multi(pa, pb, pg, n, ptr(part), ptr(row_weight), weight_period, r0, r1, d, sa0, sa1,
                                           sb0, sb1, g0, g1, gscale, _COS_EPS, ptr(loss_out), 1 if accumulate else 0,
                                           st), "vqa_neg_cos_rows_multi")
    return [ga[i] for i in range(n)] if want_grad else None


def mlm_cross_entropy(logits, label_sets, loss_out, accumulate, gscale=1.0, want_grad=True, ignore_index=-100,
                      flag=None, ws=None, rows_per_sample=0):
    """loss_out[0] (+)= gscale * sum_k CE_mean(logits, label_sets[k]); returns d(that)/d logits or None.

    ``logits`` (..., V) fp32 with dense rows; ``label_sets`` int64 (K, rows) -- K label sets over the same rows
    (K = 1 for the reference's 2-d labels, K = labels.shape[1] for its 3-d labels).
    ``rows_per_sample``: 0 -> every label set is a mean over ALL rows (``F.cross_entropy`` on the batch, the reference's
    op); L -> rows are grouped per sample (L consecutive rows) and each sample's label sets are normalised by that
    sample's own valid-label counts, i.e. the batch-1 reference's loss summed over the samples of a batched attack.
    A label outside ``[0, V)`` that is not ``ignore_index`` makes the loss NaN and sets ``VQA_FLAG_BAD_LABEL`` in
    ``flag`` (int32 device word, optional) -- torch device-asserts there; nothing is silently dropped.
    """
    dev_f32(logits, "logits", contiguous=False)
    v = logits.shape[-1]
    flat = logits.reshape(-1, v)
    if flat.stride(1) != 1:
        flat = flat.contiguous()
    rows = flat.shape[0]
    if label_sets.dtype != torch.int64 or not label_sets.is_cuda:
        raise TypeError("label_sets must be an int64 device tensor")
    label_sets = label_sets.reshape(-1, rows).contiguous()
    k = label_sets.shape[0]
    if k > lib().vqa_ce_max_label_sets():
        raise _hip.HipExtensionError("at most {} label sets per launch".format(lib().vqa_ce_max_label_sets()))
    # With a workspace the gradient buffer is the same memory in every iteration of the attack: it is zero-filled once
    # and a byte per row remembers whether the row holds a live gradient, so the rows whose labels are all ignore_index
    # (>= 90 % of a dense (B, L, V) logits tensor in the reference's workload) are never written again.
    grad = row_state = None
    if want_grad:
        if ws is not None and ws.persistent() and rows > 0:
            grad = ws.get(("ce_grad", v), (rows, v), torch.float32, logits.device, zero=True)
            row_state = ws.get(("ce_row_state", v), (rows,), torch.uint8, logits.device, zero=True)
        else:
            grad = torch.empty((rows, v), dtype=torch.float32, device=logits.device)
    groups = 1 if rows_per_sample in (0, None) or rows == 0 else -(-rows // int(rows_per_sample))
    scratch = _scratch(ws, "ce_scratch", (max(int(lib().vqa_ce_scratch_floats(k, groups)), 1),), torch.float32,
                       logits.device)
    row_loss = _scratch(ws, "ce_rows", (max(rows, 1),), torch.float32, logits.device)
    with _on(logits):
        check(lib().vqa_ce_rows(ptr(flat), flat.stride(0), ctypes.c_void_p(label_sets.data_ptr()), k, rows, v,
                                ignore_index, int(rows_per_sample or 0), ptr(scratch), ptr(grad), ptr(row_loss), gscale,
                                ptr(loss_out), 1 if accumulate else 0, ptr(flag),
                                None if row_state is None else ctypes.c_void_p(row_state.data_ptr()),
                                stream_for(logits)), "vqa_ce_rows")
    return grad.reshape(logits.shape) if want_grad else None


# ----------------------------------------------------------------------------------------- text side
class RowIndex:
    """A validated, device-resident int64 index for ``gather_rows`` (``text_emb_pick`` / ``attack_mask``): build it once
    per attack and reuse it every iteration -- no per-call host check, no host->device copy, and a device-tensor index
    is read back exactly once."""

    def __init__(self, index, length, device):
        if not isinstance(index, torch.Tensor):
            index = torch.as_tensor(list(index), dtype=torch.int64)
        host = index.detach().cpu().to(torch.int64).reshape(-1)
        if host.numel() and (int(host.max()) >= length or int(host.min()) < -length):
            raise IndexError("index out of range for dimension 1 with size {}".format(length))
        self.length = length
        self.device_index = host.to(device)

    def __len__(self):
        return self.device_index.numel()


def gather_rows(src, index):
    """src (B, L, D)[:, index] -> (B, K, D).  ``index``: list / tensor (validated on every call) or a ``RowIndex``."""
    dev_f32(src, "text gradient")
    if src.dim() != 3:
        raise ValueError("text gradient must be (B, L, D)")
    b, l, d = src.shape
    if not isinstance(index, RowIndex):
        index = RowIndex(index, l, src.device)
    elif index.length != l or index.device_index.device != src.device:
        raise ValueError("RowIndex was prepared for length {} on {}".format(index.length, index.device_index.device))
    idx = index.device_index
    k = idx.numel()
    dst = torch.empty((b, k, d), dtype=torch.float32, device=src.device)
    if k == 0 or b == 0:
        return dst
    with _on(src):
        check(lib().vqa_gather_rows(ptr(src), ctypes.c_void_p(idx.data_ptr()), ptr(dst), b, l, k, d,
                                    stream_for(src)), "vqa_gather_rows")
    return dst


def cand_dir_sim(word, pos, type_emb, gamma, beta, ln_eps, e_ori, grad, cand):
    """Scores of candidate substitutions, one fp32 per row of ``cand`` (int32 (n, 4) = sample, position, grad row, id)."""
    for name, t in (("word", word), ("pos", pos), ("type", type_emb), ("gamma", gamma), ("beta", beta),
                    ("e_ori", e_ori), ("grad", grad)):
        dev_f32(t, name)
    if cand.dtype != torch.int32 or cand.dim() != 2 or cand.shape[1] != 4 or not cand.is_cuda:
        raise TypeError("cand must be an int32 (n, 4) device tensor")
    cand = cand.contiguous()
    n = cand.shape[0]
    d = word.shape[1]
    _, l, _ = e_ori.shape
    _, k, _ = grad.shape
    out = torch.empty(n, dtype=torch.float32, device=word.device)
    with _on(word):
        check(lib().vqa_cand_dir_sim(ptr(word), ptr(pos), ptr(type_emb), ptr(gamma), ptr(beta), ln_eps, ptr(e_ori),
                                     ptr(grad), ctypes.c_void_p(cand.data_ptr()), ptr(out), n, l, k, d,
                                     stream_for(word)), "vqa_cand_dir_sim")
    return out


def embed_tokens(tables, text_ids, out=None, rows=None):
    """BERT embeddings ``LN(word[id] + type[0] + pos[p])`` on the device in one launch.

    ``text_ids`` (B, L) int64.  ``rows=None`` embeds every token into a fresh (or given) ``(B, L, D)`` tensor;
    ``rows`` = int tensor/list of ``(sample, position)`` pairs rewrites only those rows of ``out`` in place (the words
    the joint attack just substituted) using the ids currently in ``text_ids``.
    """
    word = tables["word"]
    dev_f32(word, "word")
    b, l = text_ids.shape
    d = word.shape[1]
    if out is None:
        if rows is not None:
            raise ValueError("rows=... updates an existing embedding tensor: pass out=")
        out = torch.empty(b, l, d, dtype=torch.float32, device=word.device)
    else:
        dev_f32(out, "out")
        if tuple(out.shape) != (b, l, d):
            raise ValueError("out must be ({}, {}, {})".format(b, l, d))
    ids = text_ids.to(word.device)
    if rows is None:
        s_idx = torch.arange(b, device=word.device).repeat_interleave(l)
        p_idx = torch.arange(l, device=word.device).repeat(b)
    else:
        rows = torch.as_tensor(rows, device=word.device, dtype=torch.int64).reshape(-1, 2)
        s_idx, p_idx = rows[:, 0], rows[:, 1]
    triples = torch.stack([s_idx * l + p_idx, p_idx, ids[s_idx, p_idx]], dim=1).to(torch.int32).contiguous()
    with _on(word):
        check(lib().vqa_embed_tokens(ptr(word), ptr(tables["pos"]), ptr(tables["type_emb"]), ptr(tables["gamma"]),
                                     ptr(tables["beta"]), tables["ln_eps"], ctypes.c_void_p(triples.data_ptr()),
                                     triples.shape[0], ptr(out), d, stream_for(word)), "vqa_embed_tokens")
    return out


def greedy_accept(cand, scores, ori_ids, cur_ids, table, threshold):
    """Device-side acceptance of one substitution round (``vqa_greedy_accept``); no host synchronisation.

    ``cand`` int32 (n, 4) device rows {sample, position, grad row, id}; ``scores`` fp32 (n,) their dir_sim;
    ``ori_ids`` / ``cur_ids`` int64 (B, L) device (``cur_ids`` is updated IN PLACE); ``table`` fp32 (V, E) sentence-encoder
    stand-in.  Returns ``(new_id, rank)`` int32 (B, L): the accepted id per position (-1 elsewhere) and the order in which
    the sample accepted it."""
    dev_f32(table, "table"), dev_f32(scores, "scores")
    if cand.dtype != torch.int32 or cand.dim() != 2 or cand.shape[1] != 4 or not cand.is_cuda:
        raise TypeError("cand must be an int32 (n, 4) device tensor")
    for name, t in (("ori_ids", ori_ids), ("cur_ids", cur_ids)):
        if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous() or t.dim() != 2:
            raise TypeError("{} must be a contiguous int64 (B, L) device tensor".format(name))
    b, l = cur_ids.shape
    if tuple(ori_ids.shape) != (b, l):
        raise ValueError("ori_ids / cur_ids shape mismatch")
    if l > 64:
        raise _hip.HipExtensionError("vqa_greedy_accept holds one token per lane: L <= 64, got {}".format(l))
    cand = cand.contiguous()
    n = cand.shape[0]
    # candidates grouped by sample, best dir_sim first; both sorts are stable, so ties keep their proposal order like
    # python's sorted(..., reverse=True) in the reference
    by_score = torch.argsort(scores, descending=True, stable=True)
    by_sample = torch.argsort(cand[by_score, 0], stable=True)
    order = by_score[by_sample].to(torch.int32).contiguous()
    counts = torch.bincount(cand[:, 0].to(torch.int64), minlength=b)[:b] if n else torch.zeros(b, dtype=torch.int64,
                                                                                               device=cur_ids.device)
    seg = torch.zeros(b + 1, dtype=torch.int32, device=cur_ids.device)
    seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
    new_id = torch.empty((b, l), dtype=torch.int32, device=cur_ids.device)
    rank = torch.empty((b, l), dtype=torch.int32, device=cur_ids.device)
    with _on(cur_ids):
        check(lib().vqa_greedy_accept(ctypes.c_void_p(cand.data_ptr()), ctypes.c_void_p(order.data_ptr()),
                                      ctypes.c_void_p(seg.data_ptr()), b, l, ctypes.c_void_p(ori_ids.data_ptr()),
                                      ctypes.c_void_p(cur_ids.data_ptr()), ctypes.c_void_p(new_id.data_ptr()),
                                      ctypes.c_void_p(rank.data_ptr()), ptr(table), table.shape[0], table.shape[1],
                                      float(threshold), stream_for(cur_ids)), "vqa_greedy_accept")
    return new_id, rank


# ----------------------------------------------------------------------------------------- white-box block glue
def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _rows_ok(name, t, numel):
    """The block-glue kernels index raw pointers from the row count: every operand is checked for device, dtype, layout
    and SIZE here (a short buffer would be an out-of-bounds access on the device, not an exception)."""
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError("{} must be a contiguous float32 HIP tensor".format(name))
    if t.numel() != numel:
        raise ValueError("{} has {} elements, expected {}".format(name, t.numel(), numel))


def _split_sizes(rows, d, period, split, second):
    """(elements of the segment-0 buffer, of the segment-1 buffer) of a tensor given whole or split."""
    if second is None:
        return rows * d, 0
    if period <= 0 or rows % period or not 0 <= split <= period:
        raise ValueError("a split tensor needs period > 0 dividing the row count and 0 <= split <= period")
    n0 = rows // period * split
    return n0 * d, (rows - n0) * d


def ln_fwd(x, gamma0, beta0, y0, mean, rstd, eps, r0=None, r1=None, rscale=None, x_out=None, gamma1=None, beta1=None,
           y1=None, period=0, split=0):
    """``vqa_ln_fwd`` on ``x`` (rows, D) fp32 contiguous: optional prologue ``x_out = x + rscale * r`` (r whole as ``r0`` or
    split as ``r0`` / ``r1``), then LayerNorm with the second parameter set for segment-1 rows, output whole (``y0``) or
    split (``y0`` / ``y1``).  Token layout: ``period`` rows per batch element, the first ``split`` are segment 0."""
    rows, d = x.numel() // x.shape[-1], x.shape[-1]
    _rows_ok("x", x, rows * d), _rows_ok("x_out", x_out, rows * d)
    for name, t in (("gamma0", gamma0), ("beta0", beta0), ("gamma1", gamma1), ("beta1", beta1), ("rscale", rscale)):
        _rows_ok(name, t, d)
    _rows_ok("mean", mean, rows), _rows_ok("rstd", rstd, rows)
    for name, t0, t1 in (("r", r0, r1), ("y", y0, y1)):
        n0, n1 = _split_sizes(rows, d, period, split, t1)
        _rows_ok(name + "0", t0, n0), _rows_ok(name + "1", t1, n1)
    with _on(x):
        check(lib().vqa_ln_fwd(_p(x), _p(r0), _p(r1), _p(rscale), _p(x_out), _p(gamma0), _p(beta0), _p(gamma1), _p(beta1),
                               _p(y0), _p(y1), _p(mean), _p(rstd), rows, d, period, split, eps, stream_for(x)),
              "vqa_ln_fwd")


def ln_bwd(dy0, x, mean, rstd, gamma0, dx, dy1=None, gamma1=None, g_a=None, g_inj=None, rscale=None, dr0=None, dr1=None,
           period=0, split=0):
    """``vqa_ln_bwd``: ``dx = g_a + g_inj + LayerNorm'(dy)`` and optionally ``dr = rscale * dx`` (whole or split)."""
    rows, d = x.numel() // x.shape[-1], x.shape[-1]
    for name, t in (("x", x), ("dx", dx), ("g_a", g_a), ("g_inj", g_inj)):
        _rows_ok(name, t, rows * d)
    for name, t in (("gamma0", gamma0), ("gamma1", gamma1), ("rscale", rscale)):
        _rows_ok(name, t, d)
    _rows_ok("mean", mean, rows), _rows_ok("rstd", rstd, rows)
    for name, t0, t1 in (("dy", dy0, dy1), ("dr", dr0, dr1)):
        n0, n1 = _split_sizes(rows, d, period, split, t1)
        _rows_ok(name + "0", t0, n0), _rows_ok(name + "1", t1, n1)
    with _on(x):
        check(lib().vqa_ln_bwd(_p(dy0), _p(dy1), _p(x), _p(mean), _p(rstd), _p(gamma0), _p(gamma1), _p(g_a), _p(g_inj),
                               _p(rscale), _p(dx), _p(dr0), _p(dr1), rows, d, period, split, stream_for(x)), "vqa_ln_bwd")


def gelu_fwd(h, out=None):
    dev_f32(h, "h")
    out = torch.empty_like(h) if out is None else out
    _rows_ok("out", out, h.numel())
    if h.numel():
        with _on(h):
            check(lib().vqa_gelu_fwd(_p(h), _p(out), h.numel(), stream_for(h)), "vqa_gelu_fwd")
    return out


def gelu_bwd(h, da, out=None):
    """``out = da * gelu'(h)``; ``out=None`` overwrites ``da``."""
    dev_f32(h, "h"), dev_f32(da, "da")
    out = da if out is None else out
    _rows_ok("da", da, h.numel()), _rows_ok("out", out, h.numel())
    if h.numel():
        with _on(h):
            check(lib().vqa_gelu_bwd(_p(h), _p(da), _p(out), h.numel(), stream_for(h)), "vqa_gelu_bwd")
    return out
