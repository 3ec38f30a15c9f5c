"""Embedding (T/tensorrt_llm/layers/embedding.py): [num_embeddings, embedding_dim] table, plain gather."""
from ..functional import embedding
from ..module import Module
from ..parameter import Parameter


class Embedding(Module):

    def __init__(self, num_embeddings, embedding_dim, dtype=None):
        super().__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.dtype = dtype
        self.weight = Parameter(shape=(num_embeddings, embedding_dim), dtype=dtype)

    def forward(self, x):
        return embedding(x, self.weight.value)
