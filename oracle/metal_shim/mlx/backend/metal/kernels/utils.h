// Stand-in for MLX's kernel utility header (not part of the reference tree): the reference's files only need the
// `instantiate_kernel` macro from it -- on the host the launchers instantiate the templates they call.
#pragma once
#include "metal_stdlib"
using namespace metal;
#define instantiate_kernel(name, func, ...)
