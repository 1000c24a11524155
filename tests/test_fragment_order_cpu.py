"""CPU tier: the fragment order of the batched decode step's weighted rows (csrc/qmm6.h qmm6_frag_offset) as the binding's helpers state
it -- the layout a caller of tl_decode_linear_ex(fragment_order = 1) has to produce or read (include/tinyllm_engine.h)."""

import pytest
import torch


@pytest.fixture(scope="module")
def ext(built_libs):
    import tiny_llm_ext_hip

    return tiny_llm_ext_hip


def c_offset(row: int, col: int, cols: int) -> int:
    """qmm6_frag_offset, restated from the header: [16-row block][128-column group][k-step t][lane = r + 16 c][8 elements]."""
    G, k = cols >> 7, col & 127
    return (((((row >> 4) * G + (col >> 7)) * 4 + ((k & 31) >> 3)) * 64) + (16 * (k >> 5) + (row & 15))) * 8 + (k & 7)


@pytest.mark.parametrize("rows,cols", [(1, 128), (5, 256), (16, 2560), (17, 384), (40, 1024), (64, 2560)])
def test_fragment_order_helpers_are_inverse_and_match_the_kernels_offsets(ext, rows, cols):
    x = torch.arange(rows * cols, dtype=torch.int32).reshape(rows, cols).to(torch.int16)
    f = ext.fragment_order_of(x)
    assert f.shape == ((rows + 15) // 16 * 16, cols)
    assert torch.equal(ext.rows_from_fragment_order(f, rows), x)
    flat = f.reshape(-1)
    for r in {0, rows // 2, rows - 1}:
        for c in {0, 7, 8, 31, 32, 127, cols - 128, cols - 1}:
            assert int(flat[c_offset(r, c, cols)]) == int(x[r, c]), (r, c)
    if rows % 16:  # the padding rows are zero
        assert int(f.reshape(-1, 16, cols // 128, 4, 4, 8).abs().sum()) >= 0
        back = ext.rows_from_fragment_order(f, (rows + 15) // 16 * 16)
        assert int(back[rows:].abs().sum()) == 0


def test_a_lane_fragment_is_sixteen_contiguous_bytes(ext):
    """What the consumer relies on: the 8 elements lane (r, c) needs for k-step t of group g are adjacent, and the 64 lanes of one
    (block, group, k-step) are one contiguous 1 KiB."""
    cols = 512
    for row, g, t, c in ((3, 0, 0, 0), (15, 3, 2, 1), (20, 1, 3, 3)):
        base = c_offset(row, 128 * g + 32 * c + 8 * t, cols)
        assert [c_offset(row, 128 * g + 32 * c + 8 * t + e, cols) for e in range(8)] == list(range(base, base + 8))
        block_start = c_offset(row & ~15, 128 * g + 8 * t, cols)
        lane = (row & 15) + 16 * c
        assert base == block_start + lane * 8
