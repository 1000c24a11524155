"""`extensions_ref.tiny_llm_ext_ref` -> `tiny_llm_ext_hip`, the ctypes binding of libtinyllm_hip.so with the reference
extension's Python surface (src/extensions_ref/bindings.cpp:11-65)."""
import sys as _sys

import tiny_llm_ext_hip as _impl
from tiny_llm_ext_hip import *  # noqa: F401,F403

_sys.modules[__name__] = _impl
