"""GenerationSession (T/tensorrt_llm/runtime/generation.py:141-156,413-488,782-997) over the C++ host loop.

Same objects and call sequence as the reference examples use:
    model_config = ModelConfig(num_heads=.., hidden_size=.., vocab_size=.., num_layers=.., gpt_attention_plugin=..)
    sampling_config = SamplingConfig(end_id=2, pad_id=2, num_beams=1)
    decoder = GenerationSession(model_config, engine_buffer, mapping)
    decoder.setup(batch_size, max_input_length, max_new_tokens)
    output_ids = decoder.decode(input_ids, input_lengths, sampling_config)   # int32 [batch, beams, max_in + max_new]
The per-step bookkeeping of the reference's Python loop (sequence_length = max_in + step, past_key_value_length =
[0,1] then [max_in + step, 0], masked_tokens[b, len_b:max_in] = 1, ping-pong contexts, a device->host sync per step:
generation.py:490-699, :812-821, :963) lives on the device inside tllm_session_generate."""
from dataclasses import dataclass, field

import numpy as np

from ..mapping import Mapping
from .native import NativeSession


@dataclass
class ModelConfig:
    vocab_size: int
    num_layers: int
    num_heads: int
    hidden_size: int
    gpt_attention_plugin: bool = True
    multi_query_mode: bool = False
    remove_input_padding: bool = False
    model_name: str = ''
    paged_kv_cache: bool = False
    tokens_per_block: int = 64
    use_prompt_tuning: bool = False


@dataclass
class SamplingConfig:
    end_id: int
    pad_id: int
    num_beams: int = field(default=1)
    temperature: float = field(default=1.0)
    top_k: int = field(default=1)
    top_p: float = field(default=0.0)
    length_penalty: float = field(default=1.0)
    repetition_penalty: float = field(default=1.0)
    min_length: int = field(default=1)
    presence_penalty: float = field(default=0.0)
    use_beam_hyps: bool = field(default=True)


class GenerationSession(object):

    def __init__(self, model_config: ModelConfig, engine_buffer, mapping: Mapping, debug_mode=False):
        assert isinstance(model_config, ModelConfig)
        if not model_config.gpt_attention_plugin:
            raise ValueError('the MI355X LLaMA path needs the gpt_attention plugin (RoPE lives in it)')
        if model_config.multi_query_mode:
            raise NotImplementedError('multi_query_mode is not built')
        self._model_config = model_config
        self.mapping = mapping
        self.debug_mode = debug_mode
        if mapping.tp_size > 1:
            from ..parallel import ensure_tp_communicator
            ensure_tp_communicator(mapping)
        self.runtime = NativeSession(engine=bytes(engine_buffer))
        self.runtime.vocab = model_config.vocab_size
        self.batch_size = self.max_input_length = self.max_new_tokens = 0

    @property
    def vocab_size(self):
        return self._model_config.vocab_size

    @property
    def num_layers(self):
        return self._model_config.num_layers

    @property
    def num_heads(self):
        return self._model_config.num_heads

    @property
    def hidden_size(self):
        return self._model_config.hidden_size

    def setup(self, batch_size: int, max_input_length: int, max_new_tokens: int, beam_width: int = 1):
        if not 1 <= beam_width <= 8:
            raise ValueError(f'beam_width {beam_width}: the device-side beam step takes 1 to 8 hypotheses per prompt')
        # batch_size * beam_width sequences: any number; the generation GEMVs take 8 rows per launch, more go through in slabs of 8
        # (each slab streams the weights again - build.py's default --max_batch_size 8 with beam search relies on this)
        self.batch_size, self.max_input_length, self.max_new_tokens = batch_size, max_input_length, max_new_tokens
        self.beam_width = beam_width
        self.runtime.setup(batch_size, max_input_length, max_new_tokens, beam_width)

    def decode(self, input_ids, input_lengths, sampling_config: SamplingConfig, prompt_embedding_table=None,
               tasks=None, prompt_vocab_size=None):
        """input_ids: int32 [batch, max_input_length] (torch tensor or ndarray), padded with pad_id.
        Returns int32 [batch, num_beams, max_input_length + max_new_tokens] like the reference (generation.py:991-997):
        greedy for num_beams == 1, beam search (hypotheses best first, back-tracked by gather_tree) otherwise."""
        self._check_sampling_config(sampling_config)
        if sampling_config.num_beams != getattr(self, 'beam_width', 1):
            # the reference sizes its beam buffers inside decode() from scfg.num_beams (generation.py:365-411)
            self.setup(self.batch_size, self.max_input_length, self.max_new_tokens, sampling_config.num_beams)
        is_torch = hasattr(input_ids, 'cpu')
        ids = input_ids.cpu().numpy() if is_torch else np.asarray(input_ids)
        lens = input_lengths.cpu().numpy() if hasattr(input_lengths, 'cpu') else np.asarray(input_lengths)
        assert ids.shape == (self.batch_size, self.max_input_length), 'call setup() with matching sizes first'
        out = self.runtime.generate(ids.astype(np.int32), lens.astype(np.int32), self.max_new_tokens,
                                    end_id=sampling_config.end_id, pad_id=sampling_config.pad_id)
        out = out.reshape(self.batch_size, sampling_config.num_beams, -1)
        if is_torch:
            import torch
            return torch.from_numpy(out).to(input_ids.device)
        return out

    @staticmethod
    def _check_sampling_config(scfg: SamplingConfig):
        """Only what the device-side sampler honours is accepted; a field that would silently change nothing raises
        instead (the reference feeds all of them to DynamicDecodeOp, generation.py:300-345, 949-961).
        Greedy (num_beams 1, top_k 1): temperature and top_p cannot change an arg-max and are accepted.
        Beam search: hypotheses are ranked by the raw cumulative log-probability, as the reference's runtime does without
        beam_hyps - so temperature must be 1; top_k / top_p play no part in the reference's beam search either."""
        if scfg.num_beams == 1 and scfg.top_k != 1:
            raise NotImplementedError('sampling (top_k > 1 / top_p) is not built: greedy or beam search only')
        if scfg.num_beams == 1 and not scfg.temperature > 0:
            raise ValueError('temperature must be positive')
        bad = []
        if scfg.repetition_penalty != 1.0:
            bad.append(f'repetition_penalty={scfg.repetition_penalty}')
        if scfg.presence_penalty != 0.0:
            bad.append(f'presence_penalty={scfg.presence_penalty}')
        if scfg.min_length > 1:
            bad.append(f'min_length={scfg.min_length}')
        if scfg.num_beams > 1 and scfg.temperature != 1.0:
            bad.append(f'temperature={scfg.temperature} with beam search')
        if scfg.num_beams > 1 and scfg.length_penalty != 1.0:
            bad.append(f'length_penalty={scfg.length_penalty}')
        if bad:
            raise NotImplementedError('SamplingConfig fields the device-side sampler does not honour: ' + ', '.join(bad))

    def decode_batch(self, input_ids, sampling_config: SamplingConfig):
        """list of 1-D id tensors -> pads to the longest and decodes (generation.py:770-780)."""
        lens = np.array([len(x) for x in input_ids], np.int32)
        max_len = int(lens.max())
        padded = np.full((len(input_ids), max_len), sampling_config.pad_id, np.int32)
        for i, x in enumerate(input_ids):
            padded[i, :lens[i]] = np.asarray(x.cpu() if hasattr(x, 'cpu') else x)
        return self.decode(padded, lens, sampling_config)
