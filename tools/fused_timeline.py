"""Stage clock of the fused QKV + attention launch (GPU box): where the 256 workgroups are at which microsecond.
   python tools/fused_timeline.py [context [session_key=value ...]]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import bench
from tensorrt_llm.runtime.native import NativeSession, _lib
cfg = dict(bench.LLAMA_7B)
dev = torch.device('cuda', 0)
w = bench.synth_weights(torch, cfg, 'sq', True, 1, 0, dev)
keys = {a.split('=')[0]: int(a.split('=')[1]) for a in sys.argv[2:]}  # session keys: python tools/fused_timeline.py 1024 key=value ...
s = NativeSession(dict(cfg, quant_mode=bench.QM['sq'] | bench.INT8_KV, tp_size=1, tp_rank=0, fused_timeline=1, **keys))
for k, v in w.items():
    s.set_tensor(k, v)
s.finalize()
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s.setup(1, ctx, 64)
s.fake_context(ctx, seed=1)
s.step(4)
lib = _lib()
lib.tllm_session_fused_timeline_ptr.argtypes = [ctypes.c_void_p]
lib.tllm_session_fused_timeline_ptr.restype = ctypes.c_void_p
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
names = {0: 'start', 1: 'prologue done', 2: 'q rows done', 3: 'k rows done', 6: 'v rows done', 4: 'barrier C (q in LDS)',
         5: 'barrier D (attention math)', 11: 'member 0: k, v, partials in LDS', 7: 'member 0: end (context row published)',
         8: 'row worker: O rows requested', 9: 'row worker: rows + context row in LDS', 10: 'row worker: end'}
for rnd in range(3):
    us, n = s.time_kernel('front', sweeps=4)
    torch.cuda.synchronize()
    t = np.zeros((256, 16), np.uint64)
    assert hip.hipMemcpy(t.ctypes.data, lib.tllm_session_fused_timeline_ptr(s._h), t.nbytes, 2) == 0
    t = t.astype(np.int64)
    t0 = t[:, 0].min()
    print(f'--- round {rnd}: {us:.2f} us per launch; last launch, us since the first workgroup started (min / median / max over workgroups)')
    for k in (0, 1, 2, 3, 6, 4, 5, 11, 7, 8, 9, 10):
        col = t[:32, k] if k in (7, 11) else (t[32:, k] if k in (8, 9, 10) else t[:, k])
        col = col[col > 0]
        if len(col):
            r = (col - t0) / 100.0
            print(f'   {names[k]:30s} {r.min():7.2f} {np.median(r):7.2f} {r.max():7.2f}')
    H = 32
    print(f'   extra q polls of the sweeping wave: {np.bincount(t[:, 12].astype(np.int64)).tolist()} (workgroups with 0, 1, 2, .. re-polls); '
          f'q in LDS - v rows done, by re-polls: ' + ', '.join(f'{k}: {np.median((t[:, 4] - t[:, 6])[t[:, 12] == k]) / 100.0:.2f} us' for k in sorted(set(t[:, 12].tolist()))))
    pd = t[:, 5].reshape(8, H)      # barrier D of every member (partials published right behind it)
    print(f'   partial hand-off (last member past barrier D -> member 0 has everything in LDS): median {np.median((t[:H, 11] - pd.max(0)) / 100.0):.2f} us')
for k in ('o_proj', 'gate_up', 'down'):
    print(k, s.time_kernel(k, sweeps=8))
if s.decode_form() & 4:
    lib.tllm_session_mlp_timeline_ptr.argtypes = [ctypes.c_void_p]
    lib.tllm_session_mlp_timeline_ptr.restype = ctypes.c_void_p
    mnames = {0: 'first tiles requested', 1: 'operand quantised', 2: 'row pairs done (wave 0)', 7: 'down rows requested, all pairs done', 3: 'line published',
              8: "leader: members' tags seen", 4: 'all 16 group lines seen', 5: 'intermediate row in LDS', 6: 'end'}
    for rnd in range(2):
        us, n = s.time_kernel('mlp', sweeps=4)
        torch.cuda.synchronize()
        t = np.zeros((256, 16), np.uint64)
        assert hip.hipMemcpy(t.ctypes.data, lib.tllm_session_mlp_timeline_ptr(s._h), t.nbytes, 2) == 0
        t = t.astype(np.int64)
        t0 = t[:, 0].min()
        print(f'--- one-launch MLP, round {rnd}: {us:.2f} us per launch; last launch, us since the first workgroup (min / median / max)')
        for k in (0, 1, 2, 7, 3, 8, 4, 5, 6):
            r = (t[:, k] - t0) / 100.0
            print(f'   {mnames[k]:36s} {r.min():7.2f} {np.median(r):7.2f} {r.max():7.2f}')
        print(f'   flag sweeps beyond the first: {np.bincount(t[:, 12].astype(np.int64)).tolist()}')
