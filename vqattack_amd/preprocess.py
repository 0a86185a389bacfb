"""Data formats either side of the attack: image input pipeline and ``.pt`` writer.

Input: the reference resizes every PIL image on the host with ``transforms.Resize((res, res), interpolation=Image.BICUBIC)``,
then ``ToTensor()`` and ``Normalize(0.5, 0.5)`` (``ALBEF_attack/dataset/__init__.py:17,35-39``;
``vlmo/transforms/square_transform.py:11-18``) and uploads fp32.  Here the 8-bit source is uploaded (4x fewer PCIe
bytes than fp32 at equal size) on a dedicated copy stream and resized + normalised on the device by two integer
kernels that reproduce Pillow's 8-bit bicubic bit for bit, writing straight into the image's slot of the
``(B, 3, S, S)`` batch tensor the attack consumes.

Output: the reference stores every adversarial image with ``torch.save(adv_x.cpu().detach(), '<qid>.pt')``
(``adv_attack.py:714``, shape ``(1, 3, H, W)`` fp32) -- a blocking device->host copy per sample.  ``AdvImageWriter`` keeps
that on-disk format but drains a whole batch through one pinned buffer on a side stream and serialises in a worker thread.
"""
import ctypes
import functools
import math
import os
import queue
import threading

import numpy as np
import torch

from ._hip import HipExtensionError, check, lib, ptr

_PRECISION_BITS = 32 - 8 - 2      # Pillow Resample.c: 8-bit pixels, 2 guard bits
_SUPPORT = 2.0                    # bicubic


def _cubic(x):
    # Pillow's bicubic_filter, a = -0.5 (same expression order: the taps are compared bit for bit)
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=256)
def resample_tables(in_size, out_size):
    """Pillow's ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for a full-axis bicubic resample.

    Returns ``(bounds int32 (out, 2) [first source index, tap count], taps int32 (out, ksize), ksize)``.  All
    arithmetic is IEEE double in Pillow's operation order (sequential accumulation of the weight sum included).
    """
    if in_size <= 0 or out_size <= 0:
        raise ValueError("sizes must be positive")
    scale = in_size / out_size
    filterscale = scale if scale > 1.0 else 1.0
    support = _SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    taps = np.zeros((out_size, ksize), dtype=np.int32)
    one = float(1 << _PRECISION_BITS)
    for o in range(out_size):
        center = (o + 0.5) * scale
        first = int(center - support + 0.5)
        first = first if first > 0 else 0
        last = int(center + support + 0.5)
        last = last if last < in_size else in_size
        count = last - first
        weights = []
        total = 0.0
        for i in range(count):
            w = _cubic((i + first - center + 0.5) * inv)
            weights.append(w)
            total += w
        for i, w in enumerate(weights):
            if total != 0.0:
                w = w / total
            taps[o, i] = int(w * one - 0.5) if w < 0 else int(w * one + 0.5)
        bounds[o, 0], bounds[o, 1] = first, count
    return bounds, taps, ksize


class ImagePreprocessor:
    """uint8 (H, W, 3) images of any size -> normalised fp32 ``(B, 3, size, size)`` on the device."""

    def __init__(self, size, device, mean=0.5, std=0.5):
        self.size, self.device = int(size), torch.device(device)
        if self.device.type != "cuda":
            raise HipExtensionError("ImagePreprocessor runs on an MI355X HIP device (got {})".format(self.device))
        self.mean, self.std = float(mean), float(std)
        self._tables = {}
        self._copy_stream = torch.cuda.Stream(device=self.device)

    def _device_tables(self, in_size):
        key = in_size
        if key not in self._tables:
            bounds, taps, ksize = resample_tables(in_size, self.size)
            self._tables[key] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(taps).to(self.device), ksize)
        return self._tables[key]

    def _staging(self, nbytes):
        """One of two pinned host buffers (alternating per call) of at least ``nbytes`` bytes + the event that says the
        upload that last used it has finished.  Pinning memory costs milliseconds per allocation
        (``Tensor.pin_memory()`` per image: 12 ms per 480 x 640 frame, measured) -- the buffers are allocated once and
        grown only when a batch is larger than any before."""
        slot = self._stage_turn = 1 - getattr(self, "_stage_turn", 1)
        bufs = self.__dict__.setdefault("_stage_bufs", [None, None])
        if bufs[slot] is None or bufs[slot][0].numel() < nbytes:
            bufs[slot] = [torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True), None]
        buf, busy = bufs[slot]
        if busy is not None:
            busy.synchronize()                               # the upload of two calls ago; long finished in practice
        return bufs[slot]

    def __call__(self, images, out=None):
        n, s = len(images), self.size
        if out is None:
            out = torch.empty(n, 3, s, s, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (n, 3, s, s) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp32 ({}, 3, {}, {}) tensor".format(n, s, s))
        main = torch.cuda.current_stream(self.device)
        tensors = []
        for img in images:
            t = torch.as_tensor(img)
            if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
                raise TypeError("images must be uint8 (H, W, 3), got {} {}".format(t.dtype, tuple(t.shape)))
            tensors.append(t.contiguous())
        host = [t for t in tensors if not t.is_cuda]
        staged = []
        with torch.cuda.stream(self._copy_stream):           # ONE upload per call: the batch's 8-bit pixels, back to back
            dev_all, ev = None, None
            pad16 = lambda n: (n + 15) // 16 * 16            # noqa: E731  (every image starts 16-byte aligned)
            if host:
                total = sum(pad16(t.numel()) for t in host)
                slot = self._staging(total)
                at = 0
                for t in host:
                    slot[0][at:at + t.numel()].copy_(t.reshape(-1))
                    at += pad16(t.numel())
                dev_all = torch.empty(total, dtype=torch.uint8, device=self.device)
                dev_all.copy_(slot[0][:total], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
                slot[1] = ev
            at = 0
            for t in tensors:
                if t.is_cuda:
                    e = torch.cuda.Event()
                    e.record(self._copy_stream)
                    staged.append((t, e))
                else:
                    staged.append((dev_all[at:at + t.numel()].view(t.shape), ev))
                    at += pad16(t.numel())
        st = ctypes.c_void_p(main.cuda_stream)
        with torch.cuda.device(self.device):      # the kernels launch on the CURRENT device: make it self.device
            self._resample(staged, main, st, out, s)
        if dev_all is not None:
            dev_all.record_stream(main)
        return out

    def _resample(self, staged, main, st, out, s):
        for b, (src, ev) in enumerate(staged):
            main.wait_event(ev)
            h, w, c = src.shape
            cur, cur_w = src, w
            if w != s:
                bounds, taps, ksize = self._device_tables(w)
                tmp = torch.empty(h, s, c, dtype=torch.uint8, device=self.device)
                check(lib().vqa_resize_bicubic_h_u8(ptr(src), h, w, c, ptr(taps), ptr(bounds), ksize, s, ptr(tmp), st),
                      "vqa_resize_bicubic_h_u8")
                cur, cur_w = tmp, s
            dst = ctypes.c_void_p(out.data_ptr() + b * 3 * s * s * 4)
            if h != s:
                bounds, taps, ksize = self._device_tables(h)
                check(lib().vqa_resize_bicubic_v_normalize(ptr(cur), h, cur_w, c, ptr(taps), ptr(bounds), ksize, s,
                                                           self.mean, self.std, dst, st),
                      "vqa_resize_bicubic_v_normalize")
            else:
                check(lib().vqa_resize_bicubic_v_normalize(ptr(cur), h, cur_w, c, None, None, 0, s, self.mean, self.std,
                                                           dst, st), "vqa_resize_bicubic_v_normalize")
            if src._base is None:                 # a caller's own device tensor (staged views live in one batch buffer)
                src.record_stream(main)
        return out


class AdvImageWriter:
    """Asynchronous ``<qid>.pt`` writer; files are byte-compatible with the reference's
    ``torch.save(adv_x.cpu().detach(), path)`` of a ``(1, 3, H, W)`` fp32 tensor."""

    def __init__(self, directory, device):
        self.directory = directory
        os.makedirs(directory, exist_ok=True)
        self.device = torch.device(device)
        self._stream = torch.cuda.Stream(device=self.device)
        self._queue = queue.Queue()
        self._error = None
        self._thread = threading.Thread(target=self._drain, daemon=True)
        self._thread.start()

    def _drain(self):
        while True:
            item = self._queue.get()
            if item is None:
                return
            host, event, qids = item
            try:
                event.synchronize()
                for i, q in enumerate(qids):
                    torch.save(host[i:i + 1].clone(), os.path.join(self.directory, "{}.pt".format(q)))
            except Exception as exc:      # surfaced by close(); never swallowed
                self._error = exc

    def write(self, adv_images, qids):
        """Queue a batch ``(B, 3, H, W)``; returns immediately (the copy is ordered after the producing stream)."""
        if len(qids) != adv_images.shape[0]:
            raise ValueError("one qid per image")
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._stream):
            host = torch.empty(adv_images.shape, dtype=adv_images.dtype, pin_memory=True)
            host.copy_(adv_images.detach(), non_blocking=True)
            event = torch.cuda.Event()
            event.record(self._stream)
        adv_images.record_stream(self._stream)
        self._queue.put((host, event, list(qids)))

    def close(self):
        self._queue.put(None)
        self._thread.join()
        if self._error is not None:
            raise self._error
