"""Token embeddings, dense and W4 (reference: src/tiny_llm_ref/embedding.py)."""

import torch

from ._ext import tiny_llm_ext_hip
from .basics import linear
from .quantize import QuantizedWeights, dequantize_weights, quantized_linear


class Embedding:
    def __init__(self, vocab_size: int, embedding_dim: int, weight: torch.Tensor):
        self.vocab_size = vocab_size
        self.embedding_dim = embedding_dim
        self.weight = weight

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self.weight[x.long()]

    def as_linear(self, x: torch.Tensor) -> torch.Tensor:
        return linear(x, self.weight)


class QuantizedEmbedding:
    """W4 table.  ``use_custom_kernel`` selects the fused gather+dequant kernel; without it (or without
    biases) rows are gathered and dequantised with readable torch ops (reference embedding.py:24-57)."""

    def __init__(self, vocab_size: int, embedding_dim: int, weight: QuantizedWeights, use_custom_kernel: bool = False):
        self.vocab_size = vocab_size
        self.embedding_dim = embedding_dim
        self.weight = weight
        self.use_custom_kernel = use_custom_kernel

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        w = self.weight
        if self.use_custom_kernel and w.biases is not None:
            return tiny_llm_ext_hip.quantized_embedding(
                x.to(torch.int32), w.scales, w.biases, w.weight, w.group_size, w.bits
            )
        rows = x.long()
        return dequantize_weights(
            w.weight[rows], w.scales[rows], None if w.biases is None else w.biases[rows], w.group_size, w.bits
        )

    def as_linear(self, x: torch.Tensor) -> torch.Tensor:
        return quantized_linear(x, self.weight)
