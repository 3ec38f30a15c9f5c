"""Generates tests/golden/kv_cache_manager.json by driving the REFERENCE's KVCacheManager
(/root/reference/tensorrt_llm_july-release-v1/tensorrt_llm/runtime/kv_cache_manager.py, loaded by file path: it only needs
torch) through scripted scenarios and recording, after every operation, the block-pointer table as block indices
((pointer - pool base) / block bytes; -1 = unallocated) plus the number of free blocks.  Run in the build container only; the
JSON is the fixture tests/test_kv_cache_manager.py replays against the in-repo manager."""
import importlib.util
import json
import os
import sys

import torch

REF = '/root/reference/tensorrt_llm_july-release-v1/tensorrt_llm/runtime/kv_cache_manager.py'
spec = importlib.util.spec_from_file_location('ref_kv_cache_manager', REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

# torch.Tensor.to('cuda') is not available here; the scenarios read BlocksManager.get_pointer_array (CPU) directly
SCENARIOS = [
    dict(name='greedy_growth', blocks=12, tokens_per_block=4, max_blocks_per_seq=4, beam_width=1, pools=2,
         ops=[('add', 0, 5), ('add', 1, 2), ('step', [False, False]), ('step', [False, False]), ('step', [False, False]),
              ('step', [False, True]), ('add', 2, 7), ('step', [False, False]), ('step', [True, False]),
              ('step', [False])]),
    dict(name='beam_shared_context', blocks=20, tokens_per_block=4, max_blocks_per_seq=5, beam_width=3, pools=1,
         ops=[('add', 0, 8), ('step', [False]), ('step', [False]), ('step', [False]), ('step', [False]), ('step', [False]),
              ('add', 1, 4), ('step', [False, False]), ('step', [True, False]), ('step', [False])]),
]


def table(mgr, pool_idx, pool, blocks):
    arr = mgr.blocks_manager.get_pointer_array(pool_idx)
    nbytes = pool.nelement() // (2 * blocks) * pool.element_size()
    idx = (arr - pool.data_ptr()) // nbytes
    idx[arr == 0] = -1
    return idx.tolist()


def main():
    out = []
    for sc in SCENARIOS:
        pools = [torch.zeros(sc['blocks'] * 2 * 2 * sc['tokens_per_block'] * 8, dtype=torch.float16) for _ in range(sc['pools'])]
        mgr = ref.KVCacheManager(pools, sc['blocks'], sc['tokens_per_block'], sc['max_blocks_per_seq'], sc['beam_width'])
        states = []
        for op in sc['ops']:
            if op[0] == 'add':
                mgr.add_sequence(ref.GenerationSequence(seq_idx=op[1], batch_idx=len(mgr.sequences)), op[2])
            else:
                mgr.step(list(op[1]))
            states.append(dict(tables=[table(mgr, i, p, sc['blocks']) for i, p in enumerate(pools)],
                               free=len(mgr.blocks_manager.free_blocks), lens=list(mgr.lens),
                               seq_ids=[s.get_seq_idx() for s in mgr.sequences]))
        out.append(dict(sc, states=states))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'kv_cache_manager.json')
    json.dump(out, open(path, 'w'))
    print('wrote', path, sum(len(s['states']) for s in out), 'states')


if __name__ == '__main__':
    sys.exit(main())
