"""Mapping(world_size, rank, gpus_per_node): tensor-parallel only, as the LLaMA example uses it
(T/tensorrt_llm/mapping.py:4-14; pp_size is hard-coded to 1 in generation.py:179)."""


class Mapping(object):

    def __init__(self, world_size=1, rank=0, gpus_per_node=8, tp_size=None):
        self.world_size = world_size
        self.rank = rank
        self.gpus_per_node = gpus_per_node
        self.tp_size = world_size if tp_size is None else tp_size
        self.tp_rank = rank % self.tp_size
        self.tp_group = list(range(self.tp_size))

    def has_tp(self):
        return self.tp_size > 1
