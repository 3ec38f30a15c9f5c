"""Where does the SmoothQuant engine's logit error on the TRAINED parent (tests/golden/trained_llama) come from, and what moves it?
CPU only: the shipped converter (examples/llama_quant/inmemory.py -> smoothquant / convert.generate_int8) builds the engine
tensors, bench_parity.FakeQuantSQ (the torch restatement of the SmoothQuant-static + int8-KV algorithm, pinned to the oracle by
tests/test_fakequant_checker.py) runs them teacher-forced on HF's own greedy path, errors are against the fixture's HF fp32 logits.
    python tools/sq_trained_sweep.py            # alpha sweep x {int8 KV on / off} + one-quantiser-at-a-time ablation"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant'))
import bench_parity  # noqa: E402
import inmemory  # noqa: E402
import smoothquant  # noqa: E402

FIX = os.path.join(ROOT, 'tests', 'golden', 'trained_llama')


class Ablate(bench_parity.FakeQuantSQ):
    """FakeQuantSQ with selected quantisers switched off (the value times its scale goes on un-rounded)."""

    def __init__(self, *a, skip=(), no_kv=False, **k):
        super().__init__(*a, **k)
        self.skip, self.no_kv, self._n = set(skip), no_kv, 0

    def quant(self, x16, scale):
        name = ('qkv_in', 'o_in', 'mlp_in', 'proj_in')[self._n % 4]
        self._n += 1
        if name in self.skip:
            return x16 * scale.float().reshape(())
        return super().quant(x16, scale)

    def attention(self, qkv, P, kv_oq, kv_qo):
        if self.no_kv:
            P = qkv.shape[0]  # every row a context row: keys / values never pass through the int8 cache
        return super().attention(qkv, P, kv_oq, kv_qo)


def main():
    from transformers import LlamaForCausalLM
    torch.set_num_threads(8)
    e = np.load(os.path.join(FIX, 'eval.npz'))
    cfg = json.load(open(os.path.join(FIX, 'config.json')))
    L, H = cfg['num_hidden_layers'], cfg['num_attention_heads']
    model = LlamaForCausalLM.from_pretrained(FIX).float().eval()
    calib = [torch.from_numpy(r.astype(np.int64))[None] for r in e['calib']]
    act = smoothquant.capture_activation_range(model, calib, num_samples=512)
    sd = dict(model.state_dict())
    n = e['hf_logits'].shape[0]
    NEW = e['hf_logits'].shape[1]
    hf = e['hf_logits'].astype(np.float32)

    def run(tensors, **kw):
        worst, tot, cnt = 0.0, 0.0, 0
        for i in range(n):
            P = int(e['lengths'][i])
            full = np.concatenate([e['prompts'][i, :P], e['hf_tokens'][i, :NEW - 1]]).astype(np.int64)
            fq = Ablate(torch, tensors, L, heads=H, **kw)
            lg = fq.forward(torch.from_numpy(full)[None], P, first_row=P - 1)[0].numpy()
            d = np.abs(lg - hf[i])
            worst, tot, cnt = max(worst, float(d.max())), tot + float(d.mean()), cnt + 1
        return worst, tot / cnt

    print(f'trained parent: logit scale {float(e["hf_logits_absmax"]):.1f}; errors = max / mean |dlogit| vs HF fp32, {n} prompts x {NEW} steps')
    if os.environ.get('DOWN_SWEEP'):
        for ad in (0.5, 0.8, 0.9, 0.95, 1.0):
            t = inmemory.engine_tensors(sd, L, mode='sq', act_range=act, alpha=0.5, per_channel=True, per_token=False, int8_kv=True,
                                        num_heads=H, threads=2, alpha_down=ad)
            a = run(t)
            c = run(t, skip={'qkv_in', 'o_in', 'mlp_in'}, no_kv=True)
            print(f'alpha 0.5, down_proj input alpha {ad}: static per-channel + int8 KV {a[0]:.3f} / {a[1]:.4f} | only the proj_in quantiser {c[0]:.3f} / {c[1]:.4f}')
        return
    for alpha in (0.3, 0.5, 0.6, 0.7, 0.8, 0.9):
        t = inmemory.engine_tensors(sd, L, mode='sq', act_range=act, alpha=alpha, per_channel=True, per_token=False, int8_kv=True,
                                    num_heads=H, threads=2)
        t = {k: v for k, v in t.items()}
        a = run(t)
        b = run(t, no_kv=True)
        print(f'alpha {alpha}: static per-channel + int8 KV {a[0]:.3f} / {a[1]:.4f} | without the int8 KV cache {b[0]:.3f} / {b[1]:.4f}')
        if alpha == 0.5:
            for only in ('qkv_in', 'o_in', 'mlp_in', 'proj_in'):
                skip = {'qkv_in', 'o_in', 'mlp_in', 'proj_in'} - {only}
                c = run(t, skip=skip, no_kv=True)
                print(f'    alpha 0.5, ONLY the {only} quantiser active (int8 weights everywhere, no int8 KV): {c[0]:.3f} / {c[1]:.4f}')
            c = run(t, skip={'qkv_in', 'o_in', 'mlp_in', 'proj_in'}, no_kv=True)
            print(f'    alpha 0.5, NO activation quantiser (int8 per-channel weights only): {c[0]:.3f} / {c[1]:.4f}')


if __name__ == '__main__':
    main()
