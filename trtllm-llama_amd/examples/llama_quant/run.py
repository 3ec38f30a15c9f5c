#!/usr/bin/env python3
"""run.py of the llama_quant example (T/examples/llama_quant/run.py:29-198): load config.json + the rank's engine,
tokenise (or take --input_tokens), then 55 x (setup + decode), printing the per-run latencies and
`llama-run (mean latency: X sec)` over runs 5..54."""
import argparse
import csv
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

import tensorrt_llm  # noqa: E402
from tensorrt_llm.runtime import GenerationSession, ModelConfig, SamplingConfig  # noqa: E402

from build import get_engine_name  # noqa: E402

EOS_TOKEN = 2
PAD_TOKEN = 2


def parse_arguments(args=None):
    p = argparse.ArgumentParser()
    p.add_argument('--max_output_len', type=int, required=True)
    p.add_argument('--log_level', type=str, default='error')
    p.add_argument('--engine_dir', type=str, default='llama_outputs')
    p.add_argument('--tokenizer_dir', type=str, default='.', help='Directory containing the tokenizer.model.')
    p.add_argument('--input_text', type=str, default='Born in north-east France, Soyer trained as a')
    p.add_argument('--input_tokens', dest='input_file', type=str, default=None,
                   help='CSV or Numpy file containing tokenized input. Alternative to text input.')
    p.add_argument('--output_csv', type=str, default=None)
    p.add_argument('--output_npy', type=str, default=None)
    p.add_argument('--num_beams', type=int, default=1)
    p.add_argument('--num_runs', type=int, default=55)
    return p.parse_args(args)


def load_session(engine_dir):
    with open(Path(engine_dir) / 'config.json') as f:
        config = json.load(f)
    bc = config['builder_config']
    dtype, world_size = bc['precision'], bc['tensor_parallel']
    assert world_size == tensorrt_llm.mpi_world_size(), \
        f'Engine world size ({world_size}) != Runtime world size ({tensorrt_llm.mpi_world_size()})'
    runtime_rank = tensorrt_llm.mpi_rank()
    mapping = tensorrt_llm.Mapping(world_size, runtime_rank)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(runtime_rank % max(torch.cuda.device_count(), 1))
    except ImportError:
        pass
    model_config = ModelConfig(num_heads=bc['num_heads'] // world_size, hidden_size=bc['hidden_size'] // world_size,
                               vocab_size=bc['vocab_size'], num_layers=bc['num_layers'],
                               gpt_attention_plugin=bool(config['plugin_config']['gpt_attention_plugin']),
                               multi_query_mode=bc.get('multi_query_mode', False),
                               remove_input_padding=config['plugin_config']['remove_input_padding'])
    path = Path(engine_dir) / get_engine_name('llama', dtype, world_size, runtime_rank)
    with open(path, 'rb') as f:
        engine_buffer = f.read()
    return GenerationSession(model_config, engine_buffer, mapping), runtime_rank


def generate(max_output_len, log_level='error', engine_dir='llama_outputs', input_text=None, input_file=None,
             output_csv=None, output_npy=None, tokenizer_dir=None, num_beams=1, num_runs=55):
    tensorrt_llm.logger.set_level(log_level)
    decoder, runtime_rank = load_session(engine_dir)
    sampling_config = SamplingConfig(end_id=EOS_TOKEN, pad_id=PAD_TOKEN, num_beams=num_beams)
    tokenizer = None
    if input_file is None:
        from transformers import LlamaTokenizer
        tokenizer = LlamaTokenizer.from_pretrained(tokenizer_dir, legacy=False)
    total = []
    output_ids = None
    for _ in range(num_runs):
        tensorrt_llm.profiler.reset()
        tensorrt_llm.profiler.start('llama-run')
        if input_file is None:
            ids = np.array([tokenizer.encode(input_text, add_special_tokens=False)], dtype=np.int32)
        elif input_file.endswith('.csv'):
            with open(input_file) as f:
                ids = np.array([[int(v) for v in next(csv.reader(f))]], dtype=np.int32)
        else:
            ids = np.load(input_file).astype(np.int32).reshape(1, -1)
        input_lengths = np.array([ids.shape[1]], dtype=np.int32)
        decoder.setup(ids.shape[0], int(input_lengths.max()), max_output_len)
        output_ids = decoder.decode(ids, input_lengths, sampling_config)
        if tokenizer is not None and runtime_rank == 0:
            outputs = output_ids[0, 0, ids.shape[1]:].tolist()
            tokenizer.decode(outputs)
        tensorrt_llm.profiler.stop('llama-run')
        total.append(tensorrt_llm.profiler.elapsed_time_in_sec('llama-run'))
    if runtime_rank == 0:
        out = np.asarray(output_ids)[0]
        if tokenizer is not None:
            print(f'Input: "{input_text}"')
            for b in range(out.shape[0]):
                print(f'Output: "{tokenizer.decode(out[b, ids.shape[1]:].tolist())}"')
        if output_csv is not None:
            with open(output_csv, 'w') as f:
                csv.writer(f, delimiter=',').writerows(out.tolist())
        if output_npy is not None:
            np.save(output_npy, out.astype(np.int32))
        print(total)
        warm = total[5:] if len(total) > 5 else total
        print(f'llama-run (mean latency: {np.mean(warm)} sec)')
    return output_ids


if __name__ == '__main__':
    args = parse_arguments()
    generate(**vars(args))
