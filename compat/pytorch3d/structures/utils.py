"""List / padded / packed conversions with the signatures of pytorch3d.structures.utils (independent implementation)."""
from typing import List, Optional, Sequence, Union

import torch


def list_to_padded(x: List[torch.Tensor], pad_size: Union[Sequence[int], None] = None, pad_value: float = 0.0,
                   equisized: bool = False) -> torch.Tensor:
    """[(Mi, ...)] -> (N, max M, ...) filled with pad_value.  Tensors may differ in every dimension (pytorch3d pads all of
    them); `pad_size` fixes the padded shape."""
    if equisized:
        return torch.stack(x, 0)
    if not x:
        raise ValueError("list_to_padded needs a non-empty list")
    nd = max(t.dim() for t in x)
    if any(t.dim() != nd for t in x if t.numel() > 0 or t.dim() == nd):
        x = [t if t.dim() == nd else t.reshape((0,) * nd) for t in x]
    if pad_size is None:
        shape = [max(int(t.shape[d]) for t in x) for d in range(nd)]
    else:
        if len(pad_size) != nd:
            raise ValueError("pad_size must have one entry per tensor dimension")
        shape = list(pad_size)
    out = x[0].new_full((len(x), *shape), pad_value)
    for i, t in enumerate(x):
        if t.numel() > 0:
            out[(i, *[slice(0, int(s)) for s in t.shape])] = t
    return out


def padded_to_list(x: torch.Tensor, split_size: Union[Sequence[int], Sequence[Sequence[int]], None] = None):
    """(N, M, ...) -> list of N tensors, element i cut to split_size[i] (an int = first dimension, or a shape)."""
    lst = list(x.unbind(0))
    if split_size is None:
        return lst
    if len(split_size) != len(lst):
        raise ValueError("split_size must have one entry per batch element")
    out = []
    for t, s in zip(lst, split_size):
        if isinstance(s, int):
            out.append(t[:s])
        else:
            out.append(t[tuple(slice(0, int(k)) for k in s)])
    return out


_CONSTANTS = {}   # (values, dtype, device) -> device tensor made from host values once


def device_constant(values, dtype, device):
    """a fresh device tensor holding `values` (a tuple of Python numbers): the host-to-device copy -- which waits for the stream
    when it comes from pageable memory -- is made once per distinct tuple; callers get a clone (device-side, asynchronous)"""
    device = torch.device(device)
    if device.type == "cpu":
        return torch.tensor(values, dtype=dtype)
    key = (values, dtype, str(device))
    t = _CONSTANTS.get(key)
    if t is None:
        if len(_CONSTANTS) > 4096:
            _CONSTANTS.clear()
        t = _CONSTANTS[key] = torch.tensor(values, dtype=dtype, device=device)
    return t.clone()


_TO_LIST = {}   # (sizes, device) -> item_packed_to_list_idx


def list_to_packed(x: List[torch.Tensor]):
    """[(Mi, ...)] -> packed (sum Mi, ...), num_items (N,), item_packed_first_idx (N,), item_packed_to_list_idx (sum Mi,).
    The three index tensors follow from the list's sizes, which are host integers: they are made once per distinct size
    tuple and handed out as clones (`repeat_interleave` with a device-side count waits for the GPU to learn its output size,
    and takes 60 us for 8 x 99,790 entries; train_mvr.py came here 16 times per iteration)."""
    if not x:
        raise ValueError("list_to_packed needs a non-empty list")
    dev = x[0].device
    sizes = tuple(int(t.shape[0]) for t in x)
    firsts, acc = [], 0
    for n in sizes:
        firsts.append(acc)
        acc += n
    num = device_constant(sizes, torch.int64, dev)
    first = device_constant(tuple(firsts), torch.int64, dev)
    packed = torch.cat(x, 0)
    key = (sizes, str(dev))
    to_list = _TO_LIST.get(key)
    if to_list is None:
        if len(_TO_LIST) > 64:
            _TO_LIST.clear()
        to_list = _TO_LIST[key] = torch.repeat_interleave(torch.arange(len(x), dtype=torch.int64, device=dev), num, output_size=acc)
    return packed, num, first, to_list.clone()


def packed_to_list(x: torch.Tensor, split_size: Union[list, int]):
    return list(x.split(split_size, dim=0))


def padded_to_packed(x: torch.Tensor, split_size: Optional[Sequence[int]] = None, pad_value=None) -> torch.Tensor:
    """(N, M, ...) -> (sum Mi, ...): keep the first split_size[i] rows of element i, or drop rows equal to pad_value."""
    if split_size is not None and pad_value is not None:
        raise ValueError("give split_size or pad_value, not both")
    N, M = x.shape[:2]
    if split_size is None and pad_value is None:
        return x.reshape(N * M, *x.shape[2:])
    if pad_value is not None:
        flat = x.reshape(N * M, -1)
        keep = (flat != pad_value).any(-1)
        return x.reshape(N * M, *x.shape[2:])[keep]
    return torch.cat([x[i, : int(s)] for i, s in enumerate(split_size)], 0)
