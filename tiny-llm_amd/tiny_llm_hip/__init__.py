"""tiny_llm_hip — the tiny_llm_ref operator / cache / model API on PyTorch-ROCm tensors and gfx950 HIP kernels.

Same public names as ``tiny_llm_ref`` (reference: src/tiny_llm_ref/__init__.py:1-20) for the W4A16 decode,
chunked-prefill and paged-attention path; the optional MoE / agent chapters are out of scope.
"""

from .attention import *  # noqa: F401,F403
from .basics import *  # noqa: F401,F403
from .embedding import *  # noqa: F401,F403
from .layer_norm import *  # noqa: F401,F403
from .positional_encoding import *  # noqa: F401,F403
from .quantize import *  # noqa: F401,F403
from .generate import *  # noqa: F401,F403
from .kv_cache import *  # noqa: F401,F403
from .paged_kv_cache import *  # noqa: F401,F403
from .qwen3_week1 import Qwen3ModelWeek1  # noqa: F401
from .qwen3_week2 import Qwen3ModelWeek2  # noqa: F401
from .qwen3_week3 import Qwen3ModelWeek3  # noqa: F401
from .sampler import *  # noqa: F401,F403
from .batch import *  # noqa: F401,F403
from .models import *  # noqa: F401,F403
from .moe import *  # noqa: F401,F403
from .loader import load, load_weights, TokenizerWrapper  # noqa: F401
from .week2_kernels import *  # noqa: F401,F403
from .prefix import AgentError, KvPrefixGenerator, ModelCheckpoint, PrefixReuse  # noqa: F401
