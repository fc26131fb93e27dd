"""Scan helpers. Oracle (test infrastructure) -- restates utils/functions.py."""
import torch


def prob_check(t, eps=1e-10):
    """utils/functions.py:9-17 -- values must lie in [0, 1] and be finite."""
    if torch.isnan(t).any():
        raise AssertionError("Nan in a probability tensor.")
    if not bool((t <= 1.0 + eps).all() and (t >= 0.0 - eps).all()):
        raise AssertionError("Incorrect values in a probability tensor, 0.0 <= tensor <= 1.0")


def safe_cumprod(t, dim, eps=1e-10):
    """utils/functions.py:48-66 -- cumprod(x) = exp(cumsum(log(x + eps)))."""
    if bool((t + eps < 0).any()):
        raise RuntimeError("Safe cumprod can only take non-negative tensors as input.")
    return torch.exp(torch.cumsum(torch.log(t + eps), dim))


def exclusive_cumprod(t, dim, eps=1e-10):
    """utils/functions.py:20-45 -- [1, x1, x1x2, ...]; note the leading 1 also
    goes through log(1 + eps), exactly as the reference does by prepending ones
    before safe_cumprod."""
    shape = list(t.shape)
    shape[dim] = 1
    padded = torch.cat([torch.ones(shape, dtype=t.dtype), t], dim=dim)
    return safe_cumprod(padded, dim, eps).narrow(dim, 0, t.size(dim))


def moving_sum(x, start_idx, end_idx):
    """utils/functions.py:69-125 -- out[n] = sum_{m=n-(start_idx-1)}^{n+end_idx-1} x[m]
    along the last dim of a (bsz, tgt, src) tensor, zero outside."""
    assert start_idx > 0 and end_idx > 0
    b, t, s = x.shape
    flat = x.reshape(-1, s)
    csum = torch.cat([flat.new_zeros(flat.size(0), 1), torch.cumsum(flat, 1)], 1)
    idx = torch.arange(s)
    hi = (idx + end_idx).clamp(max=s)          # exclusive upper index
    lo = (idx - (start_idx - 1)).clamp(min=0)  # inclusive lower index
    out = csum[:, hi] - csum[:, lo]
    return out.view(b, t, s)


def moving_sum_conv(x, start_idx, end_idx):
    """Same as moving_sum but with the reference's exact operation order
    (conv1d with a ones kernel, utils/functions.py:113-121); used where the
    1e-3 fp parity budget should not be spent on a different summation order."""
    assert start_idx > 0 and end_idx > 0
    b, t, s = x.shape
    w = torch.ones(1, 1, end_idx + start_idx - 1, dtype=x.dtype)
    y = torch.nn.functional.conv1d(x.reshape(-1, 1, s), w, padding=start_idx + end_idx - 1).squeeze(1)
    y = y[:, end_idx:-start_idx]
    return y.reshape(b, t, s)
