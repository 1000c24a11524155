/* placeholder: filled in with the fused decode engine ABI */
#ifndef TINYLLM_ENGINE_H
#define TINYLLM_ENGINE_H
#endif
