"""packed_to_padded / padded_to_packed with pytorch3d.ops' interface (differentiable index arithmetic)."""
import torch


def packed_to_padded(inputs, first_idxs, max_size: int):
    """(F,) or (F, D) packed values -> (N, max_size[, D]) zero-padded; first_idxs (N,) start of every element"""
    flat = inputs.dim() == 1
    x = inputs[:, None] if flat else inputs
    F_ = x.shape[0]
    N = first_idxs.shape[0]
    ends = torch.cat([first_idxs[1:], first_idxs.new_tensor([F_])])
    ar = torch.arange(max_size, device=x.device)[None]
    src = first_idxs[:, None] + ar
    valid = src < ends[:, None]
    out = x[src.clamp(max=max(F_ - 1, 0)).reshape(-1)].reshape(N, max_size, -1) if F_ > 0 else x.new_zeros(N, max_size, x.shape[1])
    out = out * valid[..., None].to(out.dtype)
    return out[..., 0] if flat else out


def padded_to_packed(inputs, first_idxs, num_inputs: int):
    """(N, max_size[, D]) -> (num_inputs[, D]); inverse of packed_to_padded"""
    flat = inputs.dim() == 2
    x = inputs[..., None] if flat else inputs
    N, M = x.shape[:2]
    ends = torch.cat([first_idxs[1:], first_idxs.new_tensor([num_inputs])])
    counts = (ends - first_idxs).clamp(min=0, max=M)
    if x.is_cuda and 0 <= int(num_inputs) <= N * M and hasattr(torch, "nonzero_static"):
        # the rows to keep, found with a FIXED output size (num_inputs is a host integer): `repeat_interleave` with device-side
        # counts and `int(counts.sum())` both wait for the GPU
        keep = (torch.arange(M, device=x.device)[None, :] < counts[:, None]).reshape(-1)
        pos = torch.nonzero_static(keep, size=int(num_inputs), fill_value=0)[:, 0]
        out = x.reshape(N * M, *x.shape[2:])[pos]
        return out[..., 0] if flat else out
    rows = torch.repeat_interleave(torch.arange(N, device=x.device), counts)
    cols = torch.arange(int(counts.sum()), device=x.device) - torch.repeat_interleave(torch.cumsum(counts, 0) - counts, counts)
    out = x[rows, cols]
    return out[..., 0] if flat else out
