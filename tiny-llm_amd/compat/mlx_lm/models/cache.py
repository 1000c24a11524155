"""`mlx_lm.models.cache.make_prompt_cache(model)` (reference benches/bench.py:320-322): one growing K/V cache per layer
for the facade's `mlx_lm.models.qwen3.Model`."""

import torch


class KVCache:
    def __init__(self):
        self.keys = None
        self.values = None
        self.offset = 0

    def update_and_fetch(self, keys, values):
        self.keys = keys if self.keys is None else torch.cat([self.keys, keys], dim=2)
        self.values = values if self.values is None else torch.cat([self.values, values], dim=2)
        self.offset = self.keys.shape[2]
        return self.keys, self.values

    @property
    def state(self):
        return self.keys, self.values


def make_prompt_cache(model, max_kv_size=None):
    return [KVCache() for _ in model.layers]
