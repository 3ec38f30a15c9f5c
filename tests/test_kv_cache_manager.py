"""Host logic of the paged KV cache (tensorrt_llm/runtime/kv_cache_manager.py) on the CPU: replay of scenarios recorded from
the reference's KVCacheManager (tests/golden/kv_cache_manager.json, made by tests/golden/make_kv_cache_manager_fixture.py) -
same block ids in the same table slots after every add_sequence / step - plus the invariants the kernels rely on."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
from tensorrt_llm.runtime.kv_cache_manager import BlocksManager, GenerationSequence, KVCacheManager  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'kv_cache_manager.json')))


def as_block_ids(mgr, pool_idx, pool, blocks):
    arr = mgr.blocks_manager.get_pointer_array(pool_idx)
    nbytes = pool.size // (2 * blocks) * pool.itemsize
    idx = (arr - pool.ctypes.data) // nbytes
    idx[arr == 0] = -1
    return idx


@pytest.mark.parametrize('sc', GOLD, ids=[s['name'] for s in GOLD])
def test_replay_of_the_reference_manager(sc):
    pools = [np.zeros(sc['blocks'] * 2 * 2 * sc['tokens_per_block'] * 8, np.float16) for _ in range(sc['pools'])]
    mgr = KVCacheManager(pools, sc['blocks'], sc['tokens_per_block'], sc['max_blocks_per_seq'], sc['beam_width'])
    for op, want in zip(sc['ops'], sc['states']):
        if op[0] == 'add':
            mgr.add_sequence(GenerationSequence(seq_idx=op[1], batch_idx=len(mgr.sequences)), op[2])
        else:
            mgr.step(list(op[1]))
        for i, pool in enumerate(pools):
            np.testing.assert_array_equal(as_block_ids(mgr, i, pool, sc['blocks']), np.array(want['tables'][i]), err_msg=str(op))
        assert mgr.blocks_manager.num_free_blocks() == want['free']
        assert mgr.lens == want['lens']
        assert [s.get_seq_idx() for s in mgr.sequences] == want['seq_ids']


def test_k_and_v_halves_and_sharing():
    blocks, T, per_tok = 6, 4, 16
    pool = np.zeros(2 * blocks * T * per_tok, np.int8)
    bm = BlocksManager([pool], blocks, max_blocks_per_seq=3, beam_width=2)
    a = GenerationSequence(0, 0)
    bm.allocate(a, share_across_beam=True)
    bm.allocate(a)
    t = bm.get_pointer_array(0)
    assert t.shape == (1, 2, 2, 3)
    blk_bytes = T * per_tok
    assert t[0, 0, 0, 0] == t[0, 1, 0, 0] == pool.ctypes.data          # shared prompt block
    assert t[0, 0, 0, 1] != t[0, 1, 0, 1]                              # private generation blocks
    np.testing.assert_array_equal(t[0, :, 1, :2] - t[0, :, 0, :2], blocks * blk_bytes)  # V half follows the K half
    assert (t[0, :, :, 2] == 0).all()
    assert bm.num_free_blocks() == 3
    bm.free(a)
    assert bm.num_free_blocks() == blocks


def test_exhaustion_and_limits():
    pool = np.zeros(2 * 2 * 64, np.float16)
    mgr = KVCacheManager([pool], blocks=2, tokens_per_block=4, max_blocks_per_seq=8)
    mgr.add_sequence(GenerationSequence(0, 0), 6)  # 7 tokens -> 2 blocks
    with pytest.raises(RuntimeError):
        mgr.add_sequence(GenerationSequence(1, 1), 1)
    with pytest.raises(ValueError):
        KVCacheManager([pool], 2, 6, 8)  # tokens_per_block must be a power of two
    with pytest.raises(ValueError):
        KVCacheManager([pool], 2, 4, 8, beam_width=2).add_sequence(GenerationSequence(0, 0), 5)
