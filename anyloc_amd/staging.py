"""Host <-> device transfers of the drop-in surface (CPU tensors in -> CPU tensors out).

The reference's scripts hand CPU tensors to ``VLAD.generate_multi`` (scripts/dino_v2_vlad.py:236-260: ``[n_img, 529, 1536]``
patch descriptors, 3.25 MB per image) and to ``get_top_k_recall`` (:372: the ``[n_db, 49 152]`` VLAD matrix, 1.97 GB at 10 000
rows) and read CPU tensors back.  SURVEY 8(b) planned "pinned buffer, async copy" for that.  Measured on the MI355X box
(tools/time_staging.py, profiles/r04_staging.log), pageable memory, one process:

    host -> device   tensor.to(device)        56 GB/s at 0.8 - 2 GB, 30 - 45 GB/s at 1 - 3 MB   (the PCIe rate)
                     pinned ring + DMA        12 GB/s (8 copy threads; 34 GB/s with exactly 4, 2.9 GB/s with one)
    device -> host   tensor.cpu()             37 GB/s at 3 MB, 6 - 7 GB/s at >= 0.8 GB (first-touch page faults of the result)
                     pinned result / ring     1 GB/s at 3 MB (a fresh pinned allocation per result), 3.6 GB/s at >= 0.8 GB

The runtime's own pageable path is already at the link rate here, and the round-4 pinned ring was slower in every case
(the host memcpy into the ring is the bottleneck), so it was removed again: these two functions are the one place the
product moves bytes between host and device, and they are plain torch copies.
"""
import torch


def to_device(t, device):
    """Tensor -> same dtype and shape on ``device`` (asynchronous on the current stream for device sources; a copy from
    pageable host memory returns once the runtime has staged it)."""
    device = torch.device(device)
    return t if t.device == device else t.to(device, non_blocking=True)


def to_host(t):
    """Device tensor -> CPU tensor (complete when the call returns)."""
    return t if t.device.type == "cpu" else t.cpu()
