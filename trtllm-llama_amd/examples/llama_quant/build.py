#!/usr/bin/env python3
"""build.py of the llama_quant example (flags: T/examples/llama_quant/build.py:40-225): QuantMode from the flags ->
LLaMAForCausalLM -> smooth_quantize / weight_only_quantize -> weights (FT dir / HF dir / random) -> plugin_config ->
trace -> Builder.build_engine -> <output_dir>/llama_<dtype>_tp<N>_rank<r>.engine + config.json."""
import argparse
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

import tensorrt_llm  # noqa: E402
from tensorrt_llm.builder import Builder  # noqa: E402
from tensorrt_llm.logger import logger  # noqa: E402
from tensorrt_llm.models import LLaMAForCausalLM, smooth_quantize, weight_only_quantize  # noqa: E402
from tensorrt_llm.network import net_guard  # noqa: E402
from tensorrt_llm.quantization import QuantMode  # noqa: E402

from weight import load_from_ft_llama, load_from_hf_llama, parse_ft_config  # noqa: E402

MODEL_NAME = 'llama'


def get_engine_name(model, dtype, tp_size, rank):
    return '{}_{}_tp{}_rank{}.engine'.format(model, dtype, tp_size, rank)


def serialize_engine(engine, path):
    logger.info(f'Serializing engine to {path}...')
    tik = time.time()
    with open(path, 'wb') as f:
        f.write(bytearray(engine))
    logger.info(f'Engine serialized. Total time: {time.time() - tik:.1f} s')


def parse_arguments(args=None):
    p = argparse.ArgumentParser()
    p.add_argument('--world_size', type=int, default=1, help='world size, only support tensor parallelism now')
    p.add_argument('--model_dir', type=str, default=None, help='FT checkpoint directory (hf_llama_convert.py output)')
    p.add_argument('--hf_model_dir', type=str, default=None, help='HF checkpoint directory (stock llama example)')
    p.add_argument('--dtype', type=str, default='float16', choices=['float32', 'bfloat16', 'float16'])
    p.add_argument('--timing_cache', type=str, default='model.cache')
    p.add_argument('--log_level', type=str, default='info')
    p.add_argument('--vocab_size', type=int, default=32000)
    p.add_argument('--n_layer', type=int, default=32)
    p.add_argument('--n_positions', type=int, default=2048)
    p.add_argument('--n_embd', type=int, default=4096)
    p.add_argument('--n_head', type=int, default=32)
    p.add_argument('--n_kv_head', type=int, default=None)
    p.add_argument('--hidden_act', type=str, default='silu')
    p.add_argument('--inter_size', type=int, default=11008)
    p.add_argument('--no_bias', action='store_false')
    p.add_argument('--max_batch_size', type=int, default=8)
    p.add_argument('--max_input_len', type=int, default=2048)
    p.add_argument('--max_output_len', type=int, default=512)
    p.add_argument('--max_beam_width', type=int, default=1)
    p.add_argument('--use_gpt_attention_plugin', nargs='?', const='float16', type=str, default=False,
                   choices=['float16', 'bfloat16', 'float32'])
    p.add_argument('--use_gemm_plugin', nargs='?', const='float16', type=str, default=False,
                   choices=['float16', 'bfloat16', 'float32'])
    p.add_argument('--parallel_build', default=False, action='store_true')
    p.add_argument('--gpus_per_node', type=int, default=8)
    p.add_argument('--builder_opt', type=int, default=None)
    p.add_argument('--output_dir', type=str, default='llama_outputs')
    p.add_argument('--multi_query_mode', default=False, action='store_true')
    p.add_argument('--remove_input_padding', default=False, action='store_true')
    p.add_argument('--use_smooth_quant', default=False, action='store_true')
    p.add_argument('--use_weight_only', default=False, action='store_true')
    p.add_argument('--weight_only_precision', type=str, default='int8', choices=['int8', 'int4'])
    p.add_argument('--per_channel', default=False, action='store_true')
    p.add_argument('--per_token', default=False, action='store_true')
    p.add_argument('--int8_kv_cache', default=False, action='store_true')
    p.add_argument('--random_seed', type=int, default=None)
    p.add_argument('--paged_kv_cache', action='store_true', default=False)
    args = p.parse_args(args)
    logger.set_level(args.log_level)
    if args.dtype == 'bfloat16':
        raise SystemExit('bfloat16 engines are not built on MI355X')
    if args.model_dir is not None:
        logger.info(f'Setting model configuration from {args.model_dir}.')
        n_embd, n_head, n_layer, n_positions, vocab_size, _, hidden_act, _, _, inter_size, mqm, dtype, *_ = \
            parse_ft_config(Path(args.model_dir) / 'config.ini')
        args.n_embd, args.n_head, args.n_layer, args.n_positions = n_embd, n_head, n_layer, n_positions
        args.vocab_size, args.hidden_act, args.inter_size, args.multi_query_mode = vocab_size, hidden_act, inter_size, mqm
    for plugin_arg in ('use_gpt_attention_plugin', 'use_gemm_plugin'):
        if not getattr(args, plugin_arg):
            logger.info(f'{plugin_arg} is not set, setting it as {args.dtype} automatically (RoPE needs the plugin).')
            setattr(args, plugin_arg, args.dtype)
    assert not (args.use_smooth_quant and args.use_weight_only), \
        'You cannot enable both SmoothQuant and INT8 weight-only together.'
    if args.use_smooth_quant:
        args.quant_mode = QuantMode.use_smooth_quant(args.per_token, args.per_channel)
    elif args.use_weight_only:
        args.quant_mode = QuantMode.use_weight_only(args.weight_only_precision == 'int4')
    else:
        args.quant_mode = QuantMode(0)
    if args.int8_kv_cache:
        args.quant_mode = args.quant_mode.set_int8_kv_cache()
    return args


def build_rank_engine(builder: Builder, builder_config, engine_name, rank, args):
    kv_dtype = args.dtype
    model = LLaMAForCausalLM(num_layers=args.n_layer, num_heads=args.n_head, hidden_size=args.n_embd,
                             vocab_size=args.vocab_size, hidden_act=args.hidden_act,
                             max_position_embeddings=args.n_positions, dtype=kv_dtype, mlp_hidden_size=args.inter_size,
                             neox_rotary_style=True, multi_query_mode=args.multi_query_mode,
                             tensor_parallel=args.world_size, tensor_parallel_group=list(range(args.world_size)),
                             quant_mode=args.quant_mode)
    if args.use_smooth_quant:
        model = smooth_quantize(model, args.quant_mode)
    elif args.use_weight_only:
        model = weight_only_quantize(model, args.quant_mode)
    if args.model_dir is not None:
        load_from_ft_llama(model, args.model_dir, rank, args.world_size, args.dtype)
    elif args.hf_model_dir is not None:
        from transformers import LlamaForCausalLM
        hf = LlamaForCausalLM.from_pretrained(args.hf_model_dir, torch_dtype='auto')
        load_from_hf_llama(model, hf, rank, args.world_size, args.dtype)
        del hf
    else:
        logger.warning('no --model_dir / --hf_model_dir: the engine is built with random (Xavier) weights')

    network = builder.create_network()
    network.trt_network.name = engine_name
    pc = network.plugin_config
    pc.set_gpt_attention_plugin(dtype=args.use_gpt_attention_plugin)
    pc.set_gemm_plugin(dtype=args.use_gemm_plugin)
    if args.use_smooth_quant:
        pc.set_smooth_quant_gemm_plugin(dtype=args.dtype)
        pc.set_rmsnorm_quantization_plugin(dtype=args.dtype)
        pc.set_quantize_tensor_plugin()
        pc.set_quantize_per_token_plugin()
    elif args.use_weight_only:
        pc.set_weight_only_quant_matmul_plugin(dtype='float16')
    if args.world_size > 1:
        pc.set_nccl_plugin(args.dtype)
    if args.remove_input_padding:
        pc.enable_remove_input_padding()
    if args.paged_kv_cache:
        pc.enable_paged_kv_cache()
    with net_guard(network):
        network.set_named_parameters(model.named_parameters())
        inputs = model.prepare_inputs(args.max_batch_size, args.max_input_len, args.max_output_len, True,
                                      args.max_beam_width)
        model(*inputs)
    builder_config.tp_rank = rank
    builder_config._values['tp_rank'] = rank
    engine = builder.build_engine(network, builder_config)
    if rank == 0:
        builder_config._values['plugin_config'] = pc
        builder.save_config(builder_config, os.path.join(args.output_dir, 'config.json'))
    return engine


def build(rank, args):
    os.makedirs(args.output_dir, exist_ok=True)
    builder = Builder()
    for cur_rank in range(args.world_size):
        if args.parallel_build and cur_rank != rank:
            continue
        int8_trt_flag = args.quant_mode.has_act_and_weight_quant() or args.quant_mode.has_int8_kv_cache()
        builder_config = builder.create_builder_config(
            name=MODEL_NAME, precision=args.dtype, timing_cache=args.timing_cache, tensor_parallel=args.world_size,
            parallel_build=args.parallel_build, num_layers=args.n_layer, num_heads=args.n_head,
            hidden_size=args.n_embd, vocab_size=args.vocab_size, hidden_act=args.hidden_act,
            max_position_embeddings=args.n_positions, max_batch_size=args.max_batch_size,
            max_input_len=args.max_input_len, max_output_len=args.max_output_len, int8=int8_trt_flag,
            opt_level=args.builder_opt, multi_query_mode=args.multi_query_mode, inter_size=args.inter_size,
            quant_mode=int(args.quant_mode))
        engine_name = get_engine_name(MODEL_NAME, args.dtype, args.world_size, cur_rank)
        engine = build_rank_engine(builder, builder_config, engine_name, cur_rank, args)
        assert engine is not None, f'Failed to build engine for rank {cur_rank}'
        serialize_engine(engine, os.path.join(args.output_dir, engine_name))


def run_build(args=None):
    args = parse_arguments(args)
    if args.random_seed is not None:
        import numpy as np
        np.random.seed(args.random_seed)
    logger.set_level(args.log_level)
    tik = time.time()
    build(0, args)
    logger.info(f'Total time of building all {args.world_size} engines: {time.time() - tik:.1f} s')


if __name__ == '__main__':
    run_build()
