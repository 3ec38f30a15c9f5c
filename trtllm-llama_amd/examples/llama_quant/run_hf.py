#!/usr/bin/env python3
"""run_hf.py (T/examples/llama_quant/run_hf.py:22-104): the HF transformers baseline, same prompt / greedy settings /
55-run timing loop, printing `llama-hf-run (mean latency: X sec)`.  Runs on the GPU through torch-ROCm when one is
visible, else on the host CPU (BASELINE.json configs[0])."""
import argparse
import time

import numpy as np
import torch
from transformers import LlamaForCausalLM, LlamaTokenizer


def parse_arguments():
    p = argparse.ArgumentParser()
    p.add_argument('--max_output_len', type=int, required=True)
    p.add_argument('--log_level', type=str, default='error')
    p.add_argument('--hf_model_location', type=str, default='./tmp/llama/7B/')
    p.add_argument('--tokenizer_dir', type=str, default=None)
    p.add_argument('--input_text', type=str, default='Born in north-east France, Soyer trained as a')
    p.add_argument('--num_beams', type=int, default=1)
    p.add_argument('--num_runs', type=int, default=55)
    p.add_argument('--device', type=str, default=None, choices=[None, 'cpu', 'cuda'])
    return p.parse_args()


def main():
    args = parse_arguments()
    device = args.device or ('cuda' if torch.cuda.is_available() else 'cpu')
    tok = LlamaTokenizer.from_pretrained(args.tokenizer_dir or args.hf_model_location, legacy=False)
    model = LlamaForCausalLM.from_pretrained(args.hf_model_location)
    model = (model.half().cuda() if device == 'cuda' else model.float()).eval()
    total = []
    text = ''
    for _ in range(args.num_runs):
        t0 = time.time()
        ids = tok.encode(args.input_text, return_tensors='pt', add_special_tokens=False).to(device)
        with torch.no_grad():
            out = model.generate(ids, max_new_tokens=args.max_output_len, top_k=1, num_beams=args.num_beams, do_sample=False,
                                 eos_token_id=2, pad_token_id=2)
        text = tok.decode(out[0, ids.shape[1]:].tolist())
        if device == 'cuda':
            torch.cuda.synchronize()
        total.append(time.time() - t0)
    print(f'Input: "{args.input_text}"')
    print(f'Output: "{text}"')
    print(total)
    warm = total[5:] if len(total) > 5 else total
    print(f'llama-hf-run (mean latency: {np.mean(warm)} sec) on {device}')


if __name__ == '__main__':
    main()
