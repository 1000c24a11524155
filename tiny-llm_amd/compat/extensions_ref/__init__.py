"""`extensions_ref` import name (reference src/extensions_ref): the native extension lives in tiny_llm_ext_hip."""
