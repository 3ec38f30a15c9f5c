from ..functional import AttentionMaskType, PositionEmbeddingType
from .attention import Attention
from .embedding import Embedding
from .linear import ColumnLinear, Linear, RowLinear
from .mlp import MLP, GatedMLP
from .normalization import RmsNorm

__all__ = ['Attention', 'AttentionMaskType', 'PositionEmbeddingType', 'ColumnLinear', 'Linear', 'RowLinear', 'Embedding',
           'MLP', 'GatedMLP', 'RmsNorm']
