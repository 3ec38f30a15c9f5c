"""Host-side bookkeeping of the paged KV cache (mirror of the classes of PY/runtime/kv_cache_manager.py, own implementation).

A pool is one layer's cache memory: 2 * blocks blocks of [num_heads, tokens_per_block, head_size] elements, all K blocks
first, then all V blocks (the K pointer of block i is base + i * block_bytes, its V pointer blocks * block_bytes further on -
kv_cache_manager.py:84-96).  The manager hands block ids to sequences; the GPTAttention plugin / generation kernels only ever
see the resulting table of device pointers [sequences, beam_width, 2, max_blocks_per_seq] (int64, shipped as int32 pairs:
gptAttentionPlugin.cpp:313-325, K/kvCacheUtils.h:34-112).

Pools are torch tensors (data_ptr / element_size / numel) or anything with the same three members.
"""
import collections
from typing import List

import numpy as np


class GenerationSequence(object):
    """Identity of a sequence (seq_idx) and its current row in the batch (batch_idx)."""

    def __init__(self, seq_idx, batch_idx):
        self.seq_idx, self.batch_idx = seq_idx, batch_idx

    def get_batch_idx(self) -> int:
        return self.batch_idx

    def get_seq_idx(self) -> int:
        return self.seq_idx

    def __eq__(self, other):
        return (getattr(other, 'seq_idx', None), getattr(other, 'batch_idx', None)) == (self.seq_idx, self.batch_idx)

    def __hash__(self):
        return hash(self.seq_idx)


def _pool_geometry(pool, blocks):
    esz = pool.element_size() if callable(getattr(pool, 'element_size', None)) else pool.itemsize
    n = pool.numel() if callable(getattr(pool, 'numel', None)) else pool.size
    base = pool.data_ptr() if callable(getattr(pool, 'data_ptr', None)) else pool.ctypes.data
    if n % (2 * blocks):
        raise ValueError(f'pool of {n} elements does not split into 2 x {blocks} blocks')
    return base, n // (2 * blocks) * esz  # base address, bytes per block


class BlocksManager(object):
    """Free list + reference counts over `blocks` block ids; per owner, per beam, the ordered list of its blocks."""

    def __init__(self, memory_pools: List, blocks: int, max_blocks_per_seq: int = 128, beam_width: int = 1):
        self.memory_pools, self.blocks = memory_pools, blocks
        self.max_blocks_per_seq, self.beam_width = max_blocks_per_seq, beam_width
        self._geom = [_pool_geometry(p, blocks) for p in memory_pools]
        self._refs = np.zeros(blocks, np.int64)
        self._free = collections.deque(range(blocks))
        self._owned = {}  # owner -> [beam][logical block] -> block id

    # -- queries
    def has_free_block(self) -> bool:
        return bool(self._free)

    def num_free_blocks(self) -> int:
        return len(self._free)

    def get_number_blocks(self, owner: GenerationSequence) -> int:
        return len(self._owned[owner][0])

    def block_ids(self, owner: GenerationSequence):
        return [list(b) for b in self._owned[owner]]

    def get_k_ptr(self, pool_idx: int, block_id: int) -> int:
        base, nbytes = self._geom[pool_idx]
        return base + block_id * nbytes

    def get_v_ptr(self, pool_idx: int, block_id: int) -> int:
        base, nbytes = self._geom[pool_idx]
        return base + (self.blocks + block_id) * nbytes

    # -- mutation
    def allocate(self, owner: GenerationSequence, share_across_beam: bool = False):
        """One more logical block for every beam of `owner`: the same physical block for all of them when shared
        (prompt blocks, read-only afterwards), one each otherwise."""
        need = 1 if share_across_beam else self.beam_width
        if len(self._free) < need:
            raise RuntimeError("Can't allocate new block for KV cache")
        rows = self._owned.setdefault(owner, [[] for _ in range(self.beam_width)])
        if len(rows[0]) >= self.max_blocks_per_seq:
            raise RuntimeError(f'sequence {owner.get_seq_idx()} already holds max_blocks_per_seq = {self.max_blocks_per_seq} blocks')
        shared = self._free.popleft() if share_across_beam else None
        for beam in range(self.beam_width):
            blk = shared if share_across_beam else self._free.popleft()
            self._refs[blk] += 1
            rows[beam].append(blk)

    def free(self, owner: GenerationSequence):
        """Drop every reference `owner` holds; blocks nobody references any more return to the free list."""
        for row in self._owned.pop(owner):
            for blk in row:
                self._refs[blk] -= 1
                if self._refs[blk] == 0:
                    self._free.append(blk)

    # -- what the kernels consume
    def get_pointer_array(self, pool_idx: int):
        """int64 [sequences, beam_width, 2, max_blocks_per_seq] (numpy): K and V block addresses, 0 where unallocated."""
        table = np.zeros((len(self._owned), self.beam_width, 2, self.max_blocks_per_seq), np.int64)
        for owner, rows in self._owned.items():
            for beam, row in enumerate(rows):
                ids = np.asarray(row, np.int64)
                base, nbytes = self._geom[pool_idx]
                table[owner.get_batch_idx(), beam, 0, :len(ids)] = base + ids * nbytes
                table[owner.get_batch_idx(), beam, 1, :len(ids)] = base + (self.blocks + ids) * nbytes
        self.pointer_array = table
        return table


class KVCacheManager(object):
    """Grows every live sequence by one token per step(), allocating a block when the next token crosses a block
    boundary, and releases finished sequences (compacting the batch indices), as PY/runtime/kv_cache_manager.py:212-290."""

    def __init__(self, memory_pools: List, blocks: int, tokens_per_block: int, max_blocks_per_seq: int, beam_width: int = 1):
        if tokens_per_block < 1 or tokens_per_block & (tokens_per_block - 1):
            raise ValueError('tokens_per_block must be a power of two')
        self.blocks_manager = BlocksManager(memory_pools, blocks, max_blocks_per_seq, beam_width)
        self.num_pools, self.tokens_per_block, self.beam_width = len(memory_pools), tokens_per_block, beam_width
        self.lens, self.sequences = [], []

    def add_sequence(self, sequence: GenerationSequence, context_len: int):
        """Blocks for the prompt plus the first generated token; prompt blocks are shared by the beams, which is only sound
        when the prompt ends on a block boundary (the reference asserts the same)."""
        if self.beam_width > 1 and context_len % self.tokens_per_block:
            raise ValueError('beam search over a paged cache needs context_len to be a multiple of tokens_per_block')
        self.sequences.append(sequence)
        self.lens.append(context_len)
        for _ in range(-(-(context_len + 1) // self.tokens_per_block)):
            self.blocks_manager.allocate(sequence, share_across_beam=True)

    def step(self, finished: List[bool]):
        keep_seq, keep_len = [], []
        for seq, n, done in zip(self.sequences, self.lens, finished):
            if done:
                self.blocks_manager.free(seq)
                continue
            if n % self.tokens_per_block == self.tokens_per_block - 1:  # token n + 1 opens a new block
                self.blocks_manager.allocate(seq)
            seq.batch_idx = len(keep_seq)
            keep_seq.append(seq)
            keep_len.append(n + 1)
        self.sequences, self.lens = keep_seq, keep_len

    def get_pointer_arrays(self, device='cuda'):
        """One table per pool (layer), on `device`, viewed as int32 [.., 2 * max_blocks_per_seq] - the form the plugin
        input takes (TensorRT had no int64 tensors; the plugin reinterprets the pairs)."""
        import torch
        out = []
        for pool in range(self.num_pools):
            t = torch.from_numpy(self.blocks_manager.get_pointer_array(pool))
            out.append((t.to(device) if device else t).view(dtype=torch.int32))
        return out
