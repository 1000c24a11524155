"""Single import point for the native extension (mirrors ``from extensions_ref import tiny_llm_ext_ref``).

The extension is mandatory: importing it without the built ``libtinyllm_hip.so`` raises ImportError.
Tests that exercise host-side logic on a CPU-only machine replace the attribute ``ext`` of the
*consumer* modules with an oracle-backed fake (tests/conftest.py); the product never does.
"""

import tiny_llm_ext_hip as tiny_llm_ext_hip  # noqa: F401  (re-exported)
