#pragma once
namespace mlx { namespace steel {
template <typename T, typename U, int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, int LDA, int LDB, typename... Rest> struct BlockMMA;
} }
