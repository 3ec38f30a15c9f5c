"""bench.py's N > 1 flow (torchrun, sharded synthetic weights, barriers, max over ranks, rank-0 JSON line) on a 1-GPU
box: two (7B layers at tp 2) or four ranks (tp 4: Ir = 2752, Vr = 8000 - BASELINE.json configs[4]) share the GPU (TLLM_TEST_SHARED_GPU=1: gloo for torch.distributed, the peer-to-peer transport for the
model's all-reduces / all-gather).  The number itself means nothing (two ranks time-slice one GPU); the run and the
finite outputs do."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('world,layers,launcher', [(2, 4, 'torchrun'), (4, 2, 'torchrun'), (2, 2, 'self')])
def test_bench_ranks_sharing_one_gpu(world, layers, launcher):
    """launcher 'torchrun': the driver's command line.  'self': plain `python bench.py --gpus N` - bench.py becomes its own launcher
    (re-executes itself under torch.distributed.run) instead of exiting (missing #2)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, TLLM_TEST_SHARED_GPU='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    head = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
            '--master-port', str(port)] if launcher == 'torchrun' else [sys.executable]
    r = subprocess.run(head + [os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', '8', '--warmup', '2', '--layers', str(layers),
                               '--no-prefill', '--no-fp16-ref', '--no-cpu-baseline'],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == world and d['config']['parallelism'] == f'tp{world}'
    assert d['config']['rccl_communicator_ranks'] is None  # the shared-GPU rig has no RCCL communicator
    assert d['step']['outputs_finite'] and d['value'] > 0 and d['steps'] == 8
    # every available transport is timed in the one run (r04): here the peer-to-peer kernel with the three-stage seam and with the
    # fused seam (no RCCL leg on the shared-GPU rig); the headline is the faster one and says which
    tr = d['config']['transports']
    assert set(tr) == {'p2p', 'p2p_fused'} and d['config']['allreduce'] in tr, tr
    assert all(v['tokens_per_s'] > 0 and v['comm_us_per_step'] > 0 for v in tr.values()), tr
    assert d['value'] == max(v['tokens_per_s'] for v in tr.values())
    # one launch per layer seam: two per layer (the logits' gather is in the head) - all-reduce, residual add, next RMSNorm and
    # the SmoothQuant quantiser of each seam in that one launch
    assert tr['p2p_fused']['comm_launches_per_step'] == 2 * layers
