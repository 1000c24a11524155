// MLX "steel" GEMM building blocks are not in the reference tree: declared only, so that the tile-GEMM kernel templates parse.
// Those kernels (and the MMA FlashAttention) are NOT built into oracle/_ref.
#pragma once
namespace mlx { namespace steel {
template <typename T, int BROWS, int BCOLS, int DST_LD, int REDUCTION_DIM, int TGP_SIZE, typename... Rest> struct BlockLoader;
} }
