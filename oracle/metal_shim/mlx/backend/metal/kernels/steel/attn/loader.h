// MLX "steel" attention loader (not in the reference tree): a do-nothing definition so that the MMA FlashAttention kernel, which
// is NOT built into oracle/_ref, parses.
#pragma once
#include <cstdlib>
namespace mlx { namespace steel {
template <typename T, int BROWS, int BCOLS, int KDST, int KSRC, int RED, int TGP, typename... Rest> struct BlockLoaderT {
    template <typename... Args> BlockLoaderT(Args &&...) { std::abort(); }
    void load_unsafe() const {}
    template <typename A> void load_safe(A) const {}
    void next() {}
};
} }
