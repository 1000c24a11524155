"""W4A16 weights and linear layers (reference: src/tiny_llm_ref/quantize.py)."""

from typing import Any

import torch

from ._ext import tiny_llm_ext_hip


class QuantizedWeights:
    """Packed 4-bit weight [K, N/8] (uint32 words, stored as int32 bits) + per-group scales/biases.

    The three ``use_*`` flags pick the kernel family exactly like the reference: rows<=8 go to the GEMV
    when ``use_simdgroup_matvec``; otherwise the matmul runs with ``use_simdgroup_matmul`` /
    ``use_split_k_matmul`` (reference quantize.py:8-46)."""

    def __init__(
        self,
        scales: torch.Tensor,
        biases: torch.Tensor | None,
        group_size: int,
        bits: int,
        weight: torch.Tensor,
        use_simdgroup_matmul: bool = False,
        use_simdgroup_matvec: bool = True,
        use_split_k_matmul: bool = False,
    ):
        self.scales = scales
        self.biases = biases
        self.group_size = group_size
        self.bits = bits
        self.weight = weight
        self.use_simdgroup_matmul = use_simdgroup_matmul
        self.use_simdgroup_matvec = use_simdgroup_matvec
        self.use_split_k_matmul = use_split_k_matmul

    @staticmethod
    def from_mlx_layer(
        mlx_layer: Any,
        use_simdgroup_matmul: bool = False,
        use_simdgroup_matvec: bool = True,
        use_split_k_matmul: bool = False,
    ) -> "QuantizedWeights":
        """Adopt a checkpoint layer object exposing weight/scales/biases/group_size/bits
        (the mlx_lm QuantizedLinear shape, reference quantize.py:29-46).  Scales and biases are kept in bf16."""
        biases = mlx_layer.biases
        return QuantizedWeights(
            scales=mlx_layer.scales.to(torch.bfloat16),
            biases=None if biases is None else biases.to(torch.bfloat16),
            group_size=mlx_layer.group_size,
            bits=mlx_layer.bits,
            weight=mlx_layer.weight,
            use_simdgroup_matmul=use_simdgroup_matmul,
            use_simdgroup_matvec=use_simdgroup_matvec,
            use_split_k_matmul=use_split_k_matmul,
        )


def _row_count(x: torch.Tensor) -> int:
    rows = 1
    for extent in x.shape[:-1]:
        rows *= extent
    return rows


def quantized_linear(x: torch.Tensor, w: QuantizedWeights, bias: torch.Tensor | None = None) -> torch.Tensor:
    """``x @ dequant(w).T (+ bias)``; at most 8 activation rows use the decode GEMV (reference quantize.py:49-90)."""
    if _row_count(x) <= 8 and w.use_simdgroup_matvec:
        y = quantized_matvec_custom(w.scales, w.biases, w.group_size, w.bits, x, w.weight, True)
    else:
        y = quantized_matmul(
            w.scales, w.biases, w.group_size, w.bits, x, w.weight, True,
            use_simdgroup=w.use_simdgroup_matmul, use_split_k=w.use_split_k_matmul,
        )
    return y if bias is None else y + bias


def dequantize_weights(
    weight: torch.Tensor, scales: torch.Tensor, biases: torch.Tensor | None, group_size: int, bits: int
) -> torch.Tensor:
    """Readable dequantisation: code * scale (+ bias) in fp32, cast to the scale dtype
    (reference quantize.py:103-121).  Nibble i of word j is element 8j+i."""
    if bits <= 0 or 32 % bits != 0:
        raise ValueError("bits must divide a 32-bit packed weight")
    per_word = 32 // bits
    words = weight.view(torch.int32) if weight.dtype != torch.int32 else weight
    shifts = torch.arange(0, 32, bits, dtype=torch.int32, device=weight.device)
    # arithmetic shift + mask recovers every field, including the top one of negative words
    codes = (words[..., None] >> shifts) & ((1 << bits) - 1)
    codes = codes.reshape(*weight.shape[:-1], weight.shape[-1] * per_word).to(torch.float32)
    wide = codes * torch.repeat_interleave(scales.to(torch.float32), group_size, dim=-1)
    if biases is not None:
        wide = wide + torch.repeat_interleave(biases.to(torch.float32), group_size, dim=-1)
    return wide.to(scales.dtype)


def dequantize_linear(mx_layer: Any) -> torch.Tensor:
    """Dense bf16 copy of a quantized checkpoint layer (Week-1 / 'kv-cache' checkpoint; reference quantize.py:93-100)."""
    return dequantize_weights(
        mx_layer.weight, mx_layer.scales, mx_layer.biases, mx_layer.group_size, mx_layer.bits
    ).to(torch.bfloat16)


def quantized_matmul(
    scales: torch.Tensor,
    biases: torch.Tensor,
    group_size: int,
    bits: int,
    a: torch.Tensor,
    b: torch.Tensor,
    transpose_b: bool = False,
    use_simdgroup: bool = False,
    use_split_k: bool = False,
) -> torch.Tensor:
    """Flatten leading dims, run the extension matmul, restore them (reference quantize.py:124-148)."""
    lead = a.shape[:-1]
    flat = a.reshape(-1, a.shape[-1])
    out = tiny_llm_ext_hip.quantized_matmul(
        scales.contiguous(), biases.contiguous(), group_size, bits, flat.contiguous(), b.contiguous(),
        transpose_b, use_simdgroup, use_split_k,
    )
    return out.reshape(*lead, -1)


def quantized_matvec_custom(
    scales: torch.Tensor,
    biases: torch.Tensor,
    group_size: int,
    bits: int,
    a: torch.Tensor,
    b: torch.Tensor,
    transpose_b: bool = False,
) -> torch.Tensor:
    """Decode GEMV entry (<= 8 rows), extension defaults use_simdgroup=True (reference quantize.py:151-173)."""
    lead = a.shape[:-1]
    flat = a.reshape(-1, a.shape[-1])
    if flat.shape[0] > 8:
        raise ValueError("quantized_matvec_custom supports at most 8 input rows")
    out = tiny_llm_ext_hip.quantized_matmul(
        scales.contiguous(), biases.contiguous(), group_size, bits, flat.contiguous(), b.contiguous(), transpose_b
    )
    return out.reshape(*lead, -1)


def quantized_matmul_vanilla(
    scales: torch.Tensor,
    biases: torch.Tensor,
    group_size: int,
    bits: int,
    a: torch.Tensor,
    b: torch.Tensor,
    transpose_b: bool = False,
) -> torch.Tensor:
    """One-thread-per-output control kernel (reference quantize.py:176-194)."""
    return quantized_matmul(scales, biases, group_size, bits, a, b, transpose_b, use_simdgroup=False)
