"""The SmoothQuant GEMM at the prefill shapes of LLaMA-7B (M = 1024) on its own, for rocprofv3:
20 launches per shape, random int8 operands.  python tools/gemm_probe.py [M] [cfg]   (cfg: tllm_gemm_set_tile_cfg id, 0 = heuristic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from tensorrt_llm.plugin import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = capi.load_library()
lib.tllm_gemm_set_tile_cfg(cfg)
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
out = bench.sq_gemm_mfma_report(torch, dev, M)
for name, r in out.items():
    print(f"{name:12s} M={r['M']} N={r['N']} K={r['K']}  best {r['us']:.1f} us  median {r['us_median']:.1f} us  {r['TOP/s']:.0f} TOP/s "
          f"= {r['frac_of_5POPs']:.3f} of 5 POP/s")
