"""ctypes view of oracle/libqwen3_oracle.so (oracle/qwen3_decode.c): a plain-C, OpenMP-threaded restatement
of one KV-cached Qwen3 W4A16 decode step.  TEST INFRASTRUCTURE ONLY: tests use it as a second checker next to
the numpy oracle, bench.py times it as the `cpu_baseline` ("port").  The product never imports this."""

from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

_PATH = Path(__file__).resolve().parent / "libqwen3_oracle.so"


class _W4(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("s", ctypes.c_void_p), ("b", ctypes.c_void_p), ("rows", ctypes.c_int),
                ("cols", ctypes.c_int)]


class _Layer(ctypes.Structure):
    _fields_ = [(n, _W4) for n in ("q", "k", "v", "o", "gate", "up", "down")] + [
        (n, ctypes.c_void_p) for n in ("input_norm", "post_norm", "q_norm", "k_norm")]


class _Config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("hidden", "layers", "heads", "kv_heads", "head_dim", "inter", "vocab",
                                            "max_ctx")] + [("rope_theta", ctypes.c_float), ("eps", ctypes.c_float)]


def available() -> bool:
    return _PATH.exists()


def _load():
    lib = ctypes.CDLL(str(_PATH))
    lib.oq_create.restype = ctypes.c_void_p
    lib.oq_create.argtypes = [ctypes.POINTER(_Config), ctypes.POINTER(_Layer), ctypes.POINTER(_W4),
                              ctypes.POINTER(_W4), ctypes.c_void_p]
    lib.oq_destroy.argtypes = [ctypes.c_void_p]
    lib.oq_reset.argtypes = [ctypes.c_void_p]
    lib.oq_context.argtypes = [ctypes.c_void_p]
    lib.oq_decode_step.restype = ctypes.c_int
    lib.oq_decode_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.oq_weight_bytes.restype = ctypes.c_double
    lib.oq_weight_bytes.argtypes = [ctypes.c_void_p]
    # float64 ground truth (qwen3_truth.c): same struct layouts
    lib.ot_create.restype = ctypes.c_void_p
    lib.ot_create.argtypes = lib.oq_create.argtypes
    lib.ot_destroy.argtypes = [ctypes.c_void_p]
    lib.ot_reset.argtypes = [ctypes.c_void_p]
    lib.ot_decode_step.restype = ctypes.c_int
    lib.ot_decode_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return lib


def _bf16_bits(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint16:
        return a
    return (np.ascontiguousarray(a, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


class COracleQwen3:
    """weights: the dict layout of tiny_oracle.make_qwen3_weights (scales/biases as bf16-representable float32 or
    raw uint16 bf16 bits)."""

    _PREFIX = "oq"
    _LOGIT_DTYPE = np.float32

    def __init__(self, cfg: dict, weights: dict, max_ctx: int = 512, threads: int | None = None):
        if threads:
            os.environ["OMP_NUM_THREADS"] = str(threads)
        self.lib = _load()
        self.cfg = cfg
        self._keep = []

        def w4(t):
            packed, s, b = t
            packed = np.ascontiguousarray(packed, dtype=np.uint32)
            s, b = _bf16_bits(s), _bf16_bits(b)
            self._keep += [packed, s, b]
            return _W4(packed.ctypes.data, s.ctypes.data, b.ctypes.data, packed.shape[0], packed.shape[1] * 8)

        def norm(a):
            bits = _bf16_bits(a)
            self._keep.append(bits)
            return bits.ctypes.data

        layers = (_Layer * cfg["num_hidden_layers"])()
        for i, lw in enumerate(weights["layers"]):
            layers[i] = _Layer(w4(lw["q"]), w4(lw["k"]), w4(lw["v"]), w4(lw["o"]), w4(lw["gate"]), w4(lw["up"]),
                               w4(lw["down"]), norm(lw["input_norm"]), norm(lw["post_norm"]), norm(lw["q_norm"]),
                               norm(lw["k_norm"]))
        embed = w4(weights["embed"])
        head = w4(weights["lm_head"]) if "lm_head" in weights else None
        c = _Config(cfg["hidden_size"], cfg["num_hidden_layers"], cfg["num_attention_heads"],
                    cfg["num_key_value_heads"], cfg["head_dim"], cfg["intermediate_size"], cfg["vocab_size"], max_ctx,
                    float(cfg["rope_theta"]), float(cfg["rms_norm_eps"]))
        create = getattr(self.lib, self._PREFIX + "_create")
        self.h = create(ctypes.byref(c), layers, ctypes.byref(embed),
                        ctypes.byref(head) if head is not None else None, norm(weights["norm"]))
        self._logits = np.zeros(cfg["vocab_size"], dtype=self._LOGIT_DTYPE)

    def reset(self) -> None:
        getattr(self.lib, self._PREFIX + "_reset")(self.h)

    def step(self, token: int):
        """Feed one token; returns (argmax id, logits[vocab]) — float32 with bf16-representable values for the bf16 port,
        unrounded float64 for the ground truth."""
        tid = getattr(self.lib, self._PREFIX + "_decode_step")(self.h, int(token), self._logits.ctypes.data)
        if tid < 0:
            raise RuntimeError("c oracle: cache full or token out of range")
        return tid, self._logits.copy()

    def weight_bytes(self) -> float:
        return float(self.lib.oq_weight_bytes(self.h))

    def close(self) -> None:
        if getattr(self, "h", None):
            getattr(self.lib, self._PREFIX + "_destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CTruthQwen3(COracleQwen3):
    """oracle/qwen3_truth.c: the same decode step in float64 with NO intermediate rounding — the ground truth the
    model-level tolerances are derived from (|HIP - truth| against |bf16 oracle - truth|)."""

    _PREFIX = "ot"
    _LOGIT_DTYPE = np.float64

    def weight_bytes(self) -> float:
        raise NotImplementedError
