"""`mlx.nn` names the reference's tests import (facade over torch; see ../../README.md)."""
import torch as _torch


def silu(x):
    return x * _torch.sigmoid(x)


def gelu(x):
    return _torch.nn.functional.gelu(x)


def relu(x):
    return _torch.relu(x)
