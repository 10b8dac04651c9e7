"""Deterministic weights / inputs shared by the golden-vector generator and the tests.

The full-size decoder has 11.7 M parameters (47 MB fp32) - far too large to commit as a
fixture.  Instead every parameter is a pure function of its *state_dict key name and shape*:
``det_tensor(name, shape)`` seeds a private CPU generator with crc32(name) and draws from it.
The generator script assigns these values to the imported reference model; the tests assign the
same values to this repo's modules (which therefore also checks key-name/shape compatibility
with the reference checkpoint layout, SURVEY.md 8(b) "Checkpoint names").

torch's CPU generator (mt19937 + the normal transform) is deterministic for a fixed torch
version; the GPU box runs the same image as the container the fixtures were made in.
"""
import zlib

import torch


def _gen(name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return g


def det_randn(name: str, shape, scale: float = 1.0) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=_gen(name), dtype=torch.float32) * scale


def det_rand(name: str, shape) -> torch.Tensor:
    return torch.rand(tuple(shape), generator=_gen(name), dtype=torch.float32)


def det_param(name: str, shape) -> torch.Tensor:
    """Value for a parameter/buffer called `name` with `shape`.

    - integer buffers (num_batches_tracked): zeros
    - running_var: in [0.5, 1.5];  running_mean: small
    - 1-D `weight` (LayerNorm / BatchNorm gamma): 1 + 0.1 n
    - other 1-D (biases): 0.05 n
    - >=2-D: n / sqrt(fan_in)   (fan_in = prod(shape[1:]) for Linear [out,in];
      for sparse-conv `kernel` [K, Cin, Cout] fan_in = K*Cin)
    """
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return 0.5 + det_rand(name, shape)
    if leaf == "running_mean":
        return det_randn(name, shape, 0.1)
    if len(shape) <= 1:
        if leaf == "weight":
            return 1.0 + det_randn(name, shape, 0.1)
        return det_randn(name, shape, 0.05)
    if leaf == "kernel":
        fan_in = 1
        for s in shape[:-1]:
            fan_in *= s
    else:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
    return det_randn(name, shape, fan_in ** -0.5)


def assign_det_weights(module: torch.nn.Module, prefix: str = "") -> None:
    """Overwrite every entry of module.state_dict() with det_param(prefix + key)."""
    sd = module.state_dict()
    new = {k: det_param(prefix + k, v.shape).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
