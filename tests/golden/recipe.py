"""Deterministic weights / inputs for the golden fixtures (shared by the generator script and
the tests; contains no reference code).  Every tensor is a function of its key name only."""
import zlib

import numpy as np
import torch


def _gen(key, salt=0):
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) + salt) % (2 ** 31))


def recipe_tensor(key, shape):
    """Value of state_dict entry `key` with logical `shape`."""
    g = _gen(key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.tensor(0, dtype=torch.long)
    if leaf == "running_mean":
        return torch.zeros(shape)
    if leaf == "running_var":
        return torch.ones(shape)
    if leaf == "weight" and len(shape) == 1:          # BatchNorm gamma
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == "weight":                               # conv / convT
        return 0.1 * torch.randn(shape, generator=g)
    return 0.1 * torch.randn(shape, generator=g)       # biases (conv or BN beta)


def recipe_state_dict(key_shapes):
    """key_shapes: {key: shape list} -> OrderedDict of tensors."""
    from collections import OrderedDict
    return OrderedDict((k, recipe_tensor(k, tuple(s))) for k, s in key_shapes.items())


def vgg_recipe():
    return recipe_state_dict({"0.weight": (64, 3, 3, 3), "0.bias": (64,),
                              "2.weight": (64, 64, 3, 3), "2.bias": (64,)})


def rand(key, shape, lo=-1.0, hi=1.0):
    return torch.rand(shape, generator=_gen(key, 7)) * (hi - lo) + lo


def randn(key, shape):
    return torch.randn(shape, generator=_gen(key, 11))


def keep_mask(key, shape):
    """0/1 dropout keep mask (p=0.5), NCHW."""
    return (torch.rand(shape, generator=_gen(key, 13)) >= 0.5).to(torch.uint8)


def sketch_indices(key, numel, n):
    """the flat positions of a (logical-layout) gradient tensor that fullsize_grad_sketch.npz keeps - all of them up to n
    elements, else n seeded draws with replacement: a function of the key only"""
    if numel <= n:
        return torch.arange(numel)
    g = torch.Generator().manual_seed(zlib.crc32(("sketch." + key).encode()) % (2 ** 31))
    return torch.randint(0, numel, (n,), generator=g)


# reduced configuration used by the module / step fixtures
SMALL = dict(ngf=8, ndf=8, n_blocks=2, n_layers_D=2, H=32, W=32, B=2)


def is_null_grad_bias(tag, key, norm):
    """Conv biases that feed an InstanceNorm directly have a mathematically zero gradient; what
    backward produces for them is rounding noise, which Adam (m/sqrt(v)) turns into +-lr steps
    whose sign differs between any two implementations (even reference vs reference with another
    summation order).  They cannot affect any output (the norm removes them), so end-state
    comparisons skip them.  tag: 'G' | 'DPB' | 'DPP'."""
    if norm != "instance" or not key.endswith(".bias"):
        return False
    if tag == "G":
        if "conv_block_stream2" in key or "conv_block_stream3" in key:
            return key.split(".")[-2] == "1"          # the block's second conv has no norm after it
        if key == "model.stream1_up.7.bias":           # head conv -> tanh
            return False
    return True
