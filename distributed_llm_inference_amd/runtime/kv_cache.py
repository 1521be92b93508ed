"""KV-cache block accounting for the scheduler.

Every stage holds a paged cache `[layers, num_blocks, Hkv, 2, 32*D]` for its own layers; block
ids are global to the pipeline (the scheduler allocates once, every stage scatters into the same
(block, offset) of its own cache). The allocator itself is native (`block_allocator.cpp`).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

from ..ops.packing import KV_BLOCK
from ..utils import native


def blocks_for(num_tokens: int) -> int:
    return (num_tokens + KV_BLOCK - 1) // KV_BLOCK


class BlockAllocator:
    def __init__(self, num_blocks: int):
        self._L = native.lib()
        self.num_blocks = int(num_blocks)
        self._h = self._L.dlia_alloc_create(self.num_blocks)
        if not self._h:
            raise MemoryError("block allocator")

    def take(self, owner: int, n: int) -> Optional[List[int]]:
        """n more blocks for `owner`, or None (nothing taken) when the pool cannot satisfy it."""
        if n == 0:
            return []
        buf = (C.c_int32 * n)()
        if self._L.dlia_alloc_take(self._h, owner, n, buf) != 0:
            return None
        return list(buf)

    def release(self, owner: int) -> int:
        return int(self._L.dlia_alloc_release(self._h, owner))

    @property
    def free(self) -> int:
        return int(self._L.dlia_alloc_free_count(self._h))

    @property
    def used(self) -> int:
        return self.num_blocks - self.free

    @property
    def peak_used(self) -> int:
        return int(self._L.dlia_alloc_peak_used(self._h))

    def owned(self, owner: int) -> int:
        return int(self._L.dlia_alloc_owned_count(self._h, owner))

    def __del__(self):
        try:
            if self._h:
                self._L.dlia_alloc_destroy(self._h)
                self._h = None
        except Exception:
            pass
