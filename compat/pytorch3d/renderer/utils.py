"""TensorProperties / convert_to_tensors_and_broadcast with pytorch3d.renderer.utils' interface (0.4: an nn.Module)."""
import copy
from typing import Any, Union

import numpy as np
import torch
import torch.nn as nn


class TensorAccessor(nn.Module):
    """view of one batch element (or a slice) of a TensorProperties object"""

    def __init__(self, class_object, index: Union[int, slice]):
        self.__dict__["class_object"] = class_object
        self.__dict__["index"] = index

    def __setattr__(self, name: str, value: Any):
        v = getattr(self.class_object, name)
        if not torch.is_tensor(v):
            raise AttributeError("Can only set values on attributes which are tensors; got %r" % type(v))
        value = torch.tensor(value, device=v.device, dtype=v.dtype) if not torch.is_tensor(value) else value
        if v.dim() > 1 and value.dim() > 1 and value.shape[1:] != v.shape[1:]:
            raise ValueError("Expected value to have shape %r; got %r" % (v.shape, value.shape))
        if v.dim() == 0 and isinstance(self.index, slice) and len(value) != len(self.index):
            raise ValueError("Expected value to have len %r; got %r" % (len(self.index), len(value)))
        self.class_object.__dict__[name][self.index] = value

    def __getattr__(self, name: str):
        if hasattr(self.class_object, name):
            return self.class_object.__dict__[name][self.index]
        raise AttributeError("Attribute %s not found on %r" % (name, self.class_object.__class__.__name__))


class TensorProperties(nn.Module):
    """Keyword arguments become batched tensor attributes, broadcast to a common batch size N."""

    def __init__(self, dtype: torch.dtype = torch.float32, device="cpu", **kwargs):
        super().__init__()
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self._N = 0
        if kwargs is not None:
            args_to_broadcast = {}
            for k, v in kwargs.items():
                if v is None or isinstance(v, (str, bool)):
                    setattr(self, k, v)
                elif isinstance(v, (tuple, list)) or torch.is_tensor(v) or isinstance(v, np.ndarray):
                    args_to_broadcast[k] = v
                else:
                    args_to_broadcast[k] = v  # python scalars
            names = list(args_to_broadcast.keys())
            values = tuple(args_to_broadcast[n] for n in names)
            if len(values) > 0:
                broadcasted = convert_to_tensors_and_broadcast(*values, device=device)
                for i, n in enumerate(names):
                    setattr(self, n, broadcasted[i])
                    if self._N == 0:
                        self._N = broadcasted[i].shape[0]

    def __len__(self) -> int:
        return self._N

    def isempty(self) -> bool:
        return self._N == 0

    def __getitem__(self, index: Union[int, slice]) -> TensorAccessor:
        if isinstance(index, (int, slice)):
            return TensorAccessor(class_object=self, index=index)
        raise ValueError("index must be an integer or slice; got %r" % type(index))

    def to(self, device="cpu"):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        for k, v in list(vars(self).items()):   # (instance attributes: dir() walks ~200 nn.Module names, 28 times per training iteration)
            if k == "device":
                setattr(self, k, device)
            if torch.is_tensor(v) and v.device != device:
                setattr(self, k, v.to(device))
        return self

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else "cuda:%d" % device)

    def clone(self, other):
        for k, v in list(vars(self).items()):   # (instance attributes: dir() walks ~200 nn.Module names, 28 times per training iteration)
            if inspect_is_plain(k, v):
                continue
            if torch.is_tensor(v):
                v_clone = v.clone()
            else:
                v_clone = copy.deepcopy(v)
            setattr(other, k, v_clone)
        return other

    def gather_props(self, batch_idx):
        for k, v in list(vars(self).items()):   # (instance attributes: dir() walks ~200 nn.Module names, 28 times per training iteration)
            if torch.is_tensor(v):
                if v.shape[0] > 1:
                    _batch_idx = batch_idx.clone()
                    idx_dims = _batch_idx.shape
                    tensor_dims = v.shape
                    if len(idx_dims) > len(tensor_dims):
                        raise ValueError("batch_idx cannot have more dimensions than the tensor")
                    if idx_dims != tensor_dims:
                        new_dims = len(tensor_dims) - len(idx_dims)
                        new_shape = idx_dims + (1,) * new_dims
                        expand_dims = (-1,) + tensor_dims[1:]
                        _batch_idx = _batch_idx.view(*new_shape).expand(expand_dims)
                    v = v.gather(0, _batch_idx)
                    setattr(self, k, v)
        return self


def inspect_is_plain(k, v) -> bool:
    """attributes TensorProperties.clone leaves alone: dunder names, methods, nn.Module internals"""
    import inspect
    return k.startswith("__") or inspect.ismethod(v) or k in ("T_destination", "dump_patches", "training", "call_super_init") \
        or (k.startswith("_") and isinstance(v, dict))


def _literal_key(x):
    """a hashable form of a Python number or a (nested) tuple / list of numbers; None for anything else"""
    if isinstance(x, (bool, int, float)):
        return (type(x).__name__, x)
    if isinstance(x, (tuple, list)):
        parts = tuple(_literal_key(v) for v in x)
        return None if any(p is None for p in parts) else ("seq",) + parts
    return None


_LITERALS = {}   # (literal, dtype, device) -> device tensor


def format_tensor(input, dtype: torch.dtype = torch.float32, device="cpu") -> torch.Tensor:
    if not torch.is_tensor(input):
        # A Python literal (shininess=64, a colour tuple, ...) becomes a device tensor through a host-to-device copy from
        # pageable memory, which on a GPU waits for the stream: 28 such copies per iteration of the reference's train_mvr.py
        # (lighting.py: convert_to_tensors_and_broadcast(..., shininess)) were 15 ms of a 41 ms iteration under the profiler.
        # The copy is made once per distinct literal; every call gets its own clone (a device-side copy, asynchronous).
        key = _literal_key(input)
        if key is not None and torch.device(device).type != "cpu":
            key = (key, dtype, str(torch.device(device)))
            cached = _LITERALS.get(key)
            if cached is None:
                if len(_LITERALS) > 4096:
                    _LITERALS.clear()
                cached = _LITERALS[key] = torch.tensor(input, dtype=dtype, device=device)
            input = cached.clone()
        else:
            input = torch.tensor(input, dtype=dtype, device=device)
    if input.dim() == 0:
        input = input.view(1)
    if input.device != torch.device(device):
        input = input.to(device=device)
    return input


def convert_to_tensors_and_broadcast(*args, dtype: torch.dtype = torch.float32, device="cpu"):
    """tensors / scalars / tuples -> tensors whose batch (first) dimension is broadcast to the largest one"""
    args_1d = [format_tensor(c, dtype, device) for c in args]
    sizes = [c.shape[0] for c in args_1d]
    N = max(sizes)
    out = []
    for c in args_1d:
        if c.shape[0] != 1 and c.shape[0] != N:
            raise ValueError("Got non-broadcastable sizes %r" % sizes)
        expand_sizes = (N,) + (-1,) * len(c.shape[1:])
        out.append(c.expand(*expand_sizes))
    return out
