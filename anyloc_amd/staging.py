"""Host <-> device staging through pinned buffers ("CPU tensors in -> pinned buffer, async copy").

The reference's scripts hand CPU tensors to ``VLAD.generate_multi`` (scripts/dino_v2_vlad.py:236-260: ``[n_img, 529, 1536]``
patch descriptors, 3.25 MB per image) and to ``get_top_k_recall`` (:372: the ``[n_db, 49 152]`` VLAD matrix, 1.97 GB at 10 000
rows) and read CPU tensors back.  A ``tensor.to(device)`` from pageable memory is a synchronous copy the runtime stages
through its own small bounce buffer; here the host side of every transfer is a page-locked buffer and the DMA is
asynchronous:

* ``to_device``: the source is copied chunk by chunk (32 MiB) into a ring of pinned buffers by several host threads
  while the previous chunk's DMA is in flight;
* ``to_host``: results up to 64 MiB are written by ONE asynchronous DMA straight into a pinned result tensor (torch's
  caching host allocator recycles the pages); larger ones go through the ring into an ordinary tensor, the host memcpy of
  chunk i running under the DMA of chunk i + 1.

No arithmetic happens here; the module only moves bytes.
"""
from concurrent.futures import ThreadPoolExecutor

import torch

CHUNK_BYTES = 32 << 20
RING = 3
PINNED_RESULT_MAX = 64 << 20

COPY_THREADS = 8               # host cores that fill / drain a pinned chunk together
_rings = {}
_pool = None


def _host_copy(dst, src):
    """dst[:] = src for two flat uint8 CPU tensors.  One ``Tensor.copy_`` is a single-threaded memcpy (~8 GB/s from pageable
    memory, measured on the MI355X host: profiles/r04_staging.log); COPY_THREADS slices run concurrently instead (``copy_``
    releases the GIL), which is what lets the PCIe DMA -- not the host -- bound a large transfer."""
    global _pool
    n = dst.numel()
    if n < (4 << 20):
        dst.copy_(src)
        return
    if _pool is None:
        _pool = ThreadPoolExecutor(max_workers=COPY_THREADS, thread_name_prefix="anyloc-staging")
    step = -(-n // COPY_THREADS)
    step = (step + 4095) // 4096 * 4096
    list(_pool.map(lambda o: dst[o:o + step].copy_(src[o:o + step]), range(0, n, step)))


class _Ring:
    def __init__(self):
        self.bufs = [torch.empty(CHUNK_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(RING)]
        self.events = [None] * RING
        self.i = 0

    def next(self):
        """-> (slot index, pinned buffer); waits until the DMA that last used the buffer has finished."""
        i = self.i
        self.i = (i + 1) % RING
        if self.events[i] is not None:
            self.events[i].synchronize()
        return i, self.bufs[i]

    def mark(self, i):
        ev = self.events[i]
        if ev is None:
            ev = self.events[i] = torch.cuda.Event()
        ev.record()


def _ring(device):
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream)
    r = _rings.get(key)
    if r is None:
        r = _rings[key] = _Ring()
    return r


def release():
    _rings.clear()


def to_device(t, device):
    """CPU (or other-device) tensor -> tensor of the same dtype and shape on ``device``; CPU sources travel through the
    pinned ring, asynchronously on the current stream (later kernels on that stream see the data)."""
    device = torch.device(device)
    if t.device.type != "cpu":
        return t if t.device == device else t.to(device, non_blocking=True)
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=t.dtype, device=device)
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return out
    if t.is_pinned():
        out.copy_(t, non_blocking=True)
        return out
    src = t.reshape(-1).view(torch.uint8)
    dst = out.reshape(-1).view(torch.uint8)
    ring = _ring(device)
    for off in range(0, nbytes, CHUNK_BYTES):
        n = min(CHUNK_BYTES, nbytes - off)
        i, buf = ring.next()
        _host_copy(buf[:n], src[off:off + n])
        dst[off:off + n].copy_(buf[:n], non_blocking=True)
        ring.mark(i)
    return out


def to_host(t):
    """Device tensor -> CPU tensor (complete when the call returns)."""
    if t.device.type == "cpu":
        return t
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes <= PINNED_RESULT_MAX:
        out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        if nbytes:
            out.copy_(t, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return out
    out = torch.empty(t.shape, dtype=t.dtype)
    src = t.reshape(-1).view(torch.uint8)
    dst = out.reshape(-1).view(torch.uint8)
    ring = _ring(t.device)
    pending = []                                        # (slot, offset, bytes): DMAs issued, host copy outstanding
    def drain(limit):
        while len(pending) > limit:
            i, off, n = pending.pop(0)
            ring.events[i].synchronize()
            _host_copy(dst[off:off + n], ring.bufs[i][:n])
            ring.events[i] = None                       # the buffer is free as soon as the host copy is done
    for off in range(0, nbytes, CHUNK_BYTES):
        n = min(CHUNK_BYTES, nbytes - off)
        drain(RING - 1)
        i, buf = ring.next()
        buf[:n].copy_(src[off:off + n], non_blocking=True)
        ring.mark(i)
        pending.append((i, off, n))
    drain(0)
    return out
