"""Readable rotary position embedding (reference: src/tiny_llm_ref/positional_encoding.py:4-66)."""

import torch


class RoPE:
    """Table-based RoPE over x=[B, L, H, D].  ``traditional`` rotates interleaved pairs (2i, 2i+1); the
    default rotates (i, i + dims/2).  ``offset`` is a slice (shared) or one slice per batch row."""

    def __init__(self, dims: int, seq_len: int, base: int = 10000, traditional: bool = False):
        if dims % 2:
            raise AssertionError("dims must be even")
        self.dims = dims
        self.seq_len = seq_len
        self.base = base
        self.traditional = traditional
        self.half_dims = dims // 2
        exponent = torch.arange(self.half_dims, dtype=torch.float32) / self.half_dims
        angles = torch.outer(torch.arange(seq_len, dtype=torch.float32), torch.pow(float(base), -exponent))
        self.cos_freqs = torch.cos(angles)
        self.sin_freqs = torch.sin(angles)

    def _tables(self, device, positions):
        if self.cos_freqs.device != device:
            self.cos_freqs = self.cos_freqs.to(device)
            self.sin_freqs = self.sin_freqs.to(device)
        return self.cos_freqs[positions], self.sin_freqs[positions]

    def __call__(self, x: torch.Tensor, offset: list[slice] | slice | None = None) -> torch.Tensor:
        B, L, H, D = x.shape
        if offset is None:
            positions = torch.arange(L, device=x.device)[None]
        elif isinstance(offset, slice):
            assert offset.stop - offset.start == L, f"offset must be of length {L}"
            positions = torch.arange(offset.start, offset.stop, device=x.device)[None]
        else:
            assert len(offset) == B, f"offsets must have the same length as batch size {B}"
            for one in offset:
                assert one.stop - one.start == L, f"offset must be of length {L}"
            positions = torch.tensor([list(range(o.start, o.stop)) for o in offset], device=x.device)
        cos, sin = self._tables(x.device, positions)  # [1|B, L, half]
        cos = cos[:, :, None, :]
        sin = sin[:, :, None, :]
        if self.traditional:
            pairs = x[..., : self.dims].reshape(B, L, H, self.half_dims, 2)
            first, second = pairs[..., 0], pairs[..., 1]
        else:
            first, second = x[..., : self.half_dims], x[..., self.half_dims : self.dims]
        real = first * cos - second * sin
        imag = second * cos + first * sin
        if self.traditional:
            rotated = torch.stack([real, imag], dim=-1).reshape(B, L, H, self.dims)
        else:
            rotated = torch.cat([real, imag], dim=-1)
        if self.dims < D:
            rotated = torch.cat([rotated.to(x.dtype), x[..., self.dims :]], dim=-1)
        return rotated.to(x.dtype)
