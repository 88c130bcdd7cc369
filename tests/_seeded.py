"""Deterministic, construction-order-independent tensor fill keyed by state-dict name.

Used by tools/make_golden.py (on the *reference's* modules, in the build
container) and by the tests (on this package's modules) so that both sides
hold bit-identical weights without shipping 137 M parameters as a fixture.
"""
import zlib

import torch


def _gen(key):
    g = torch.Generator()
    g.manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    return g


def seeded_value(key, t):
    """Value for state-dict entry `key` with the shape/dtype of tensor t (CPU fp32)."""
    g = _gen(key)
    shape = tuple(t.shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=t.dtype)
    if key.endswith("running_var"):
        return torch.rand(shape, generator=g) + 0.5
    if key.endswith("running_mean"):
        return torch.randn(shape, generator=g) * 0.1
    if t.dim() == 1 and key.endswith(".weight"):  # BN gamma
        return torch.rand(shape, generator=g) * 0.5 + 0.5
    if t.dim() == 1:  # biases (BN beta, conv bias)
        return torch.randn(shape, generator=g) * 0.1
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5


def seeded_fill_(module, prefix=""):
    """Overwrite every parameter/buffer of `module` in place; key = prefix + state-dict name."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            t.copy_(seeded_value(prefix + name, t).to(t.dtype))
    return module


def seeded_input(key, shape, scale=1.0):
    return torch.randn(shape, generator=_gen(key)) * scale


def sample_idx(numel, k=4096):
    if numel <= k:
        return torch.arange(numel)
    return torch.linspace(0, numel - 1, k).long().clamp_(max=numel - 1)   # (fp32 linspace: numel - 1 > 2^24 can round up to numel)


def summarize(t, k=4096):
    """Fixed subsample + moments of an NCHW-ordered tensor (what the goldens store)."""
    f = t.detach().float().contiguous().reshape(-1)
    d = f.double()
    return {
        "shape": torch.tensor(list(t.shape)),
        "sample": f[sample_idx(f.numel(), k)].clone(),
        "sum": d.sum().reshape(1),
        "sumsq": (d * d).sum().reshape(1),
    }
