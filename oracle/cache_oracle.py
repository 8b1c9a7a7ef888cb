"""numpy restatement of the KV-cache maintenance ops (TEST INFRASTRUCTURE, see oracle/__init__.py).

All three are pure byte/index work and must be reproduced bit-exactly.  Arrays are
``uint16`` bit patterns (the element type is irrelevant: the reference itself copies
f16 and bf16 as int16, csrc/kernels/cache_manager.cu:40-41).
"""
import numpy as np


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping):
    """csrc/kernels/cache_manager.cu:139-170.  key,value ``[T,hk,d]`` (any row stride),
    caches ``[nb,page,hk,d]`` modified in place; ``slot < 0`` = padding token, skipped;
    ``dst = cache[slot // page][slot % page]``."""
    page = key_cache.shape[1]
    for t, s in enumerate(np.asarray(slot_mapping, np.int64)):
        if s < 0:
            continue
        key_cache[s // page, s % page] = key[t]
        value_cache[s // page, s % page] = value[t]


def copy_blocks(key_caches, value_caches, block_mapping):
    """csrc/kernels/cache_manager.cu:15-37.  For every layer and pair ``(src,dst)``:
    ``cache[dst] = cache[src]`` for K and V.  The CUDA grid runs all pairs concurrently,
    so the defined behaviour is for distinct dsts that are not srcs of another pair
    (csrc/tests/cache_manager_tests.rs:255-256); the oracle reads every source from
    the *pre-copy* state to make that explicit."""
    block_mapping = np.asarray(block_mapping, np.int64).reshape(-1, 2)
    for cache in list(key_caches) + list(value_caches):
        before = cache.copy()
        for s, d in block_mapping:
            cache[d] = before[s]


def swap_blocks(src, dst, block_mapping):
    """csrc/src/cache_manager.rs:18-128: whole-page copies ``dst[d] = src[s]`` with byte
    offsets ``blk * page_bytes`` (:23-27,38-45); src/dst ``[nb,page,hk,d]``."""
    items = block_mapping.items() if hasattr(block_mapping, "items") else block_mapping
    for s, d in items:
        dst[d] = src[s]


def slot_mapping_for(block_table_row, start, stop, page):
    """backends/vllm/src/worker.rs:392-401: ``slot = block_table[i // page] * page + i % page``."""
    i = np.arange(start, stop)
    return np.asarray(block_table_row, np.int64)[i // page] * page + i % page
