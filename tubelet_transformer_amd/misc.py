"""Padded-batch container and small host helpers of the TubeR model API.

Mirrors the hot-path subset of the reference's ``utils/misc.py``:
``NestedTensor`` (``utils/misc.py:405-425``), ``nested_tensor_from_tensor_list``
(``:367-402``), ``collate_fn`` (``:279-282``), ``accuracy`` (``:521-539``) and
``accuracy_sigmoid`` (``:497-518``).  Everything else in that file is dead code
in the reference and is not reproduced.
"""
from typing import List, Optional

import torch
from torch import Tensor


class NestedTensor(object):
    """A batch of clips padded to a common size plus the padding mask (True = padding)."""

    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        mask = self.mask.to(device) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device), mask)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    """Pad a list of (C,H,W) images or (C,T,H,W) clips to the batch maximum.

    Also accepts a batched tensor (iterated along dim 0), as the reference does
    when ``DETR.forward`` is handed a plain 5-D tensor (``models/tuber_ava.py:112-113``).
    """
    first = tensor_list[0]
    if first.ndim not in (3, 4):
        raise ValueError("not supported")
    shapes = [list(t.shape) for t in tensor_list]
    max_size = [max(s[d] for s in shapes) for d in range(first.ndim)]
    b = len(shapes)
    out = torch.zeros([b] + max_size, dtype=first.dtype, device=first.device)
    mask = torch.ones((b, max_size[-2], max_size[-1]), dtype=torch.bool, device=first.device)
    for i, t in enumerate(tensor_list):
        sl = tuple(slice(0, n) for n in t.shape)
        out[i][sl].copy_(t)
        mask[i, : t.shape[-2], : t.shape[-1]] = False
    return NestedTensor(out, mask)


def collate_fn(batch):
    batch = list(zip(*batch))
    batch[0] = nested_tensor_from_tensor_list(batch[0])
    return tuple(batch)


@torch.no_grad()
def accuracy_sigmoid(output, target):
    """Exact-set accuracy used only for logging (``utils/misc.py:497-518``)."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    hits = 0
    for n in range(target.shape[0]):
        labels = target[n].nonzero().flatten()
        k = labels.numel()
        pred = output[n].topk(k, 0, True, True)[1]
        if set(labels.tolist()) == set(pred.tolist()):
            hits += 1
    return [hits * (100.0 / target.shape[0])]


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """Top-k precision used only for logging (``utils/misc.py:521-539``)."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0).mul_(100.0 / target.size(0)) for k in topk]


def is_dist_avail_and_initialized():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    import torch.distributed as dist
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    import torch.distributed as dist
    return dist.get_rank() if is_dist_avail_and_initialized() else 0
