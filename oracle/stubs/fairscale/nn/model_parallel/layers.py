import torch.nn as nn


def _init(weight, init_method):
    if init_method is not None:
        init_method(weight)


class ColumnParallelLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, gather_output=True, init_method=None, **kw):
        super().__init__(in_features, out_features, bias=bias)
        _init(self.weight, init_method)
        if self.bias is not None:
            nn.init.zeros_(self.bias)


class RowParallelLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, input_is_parallel=False, init_method=None, **kw):
        super().__init__(in_features, out_features, bias=bias)
        _init(self.weight, init_method)
        if self.bias is not None:
            nn.init.zeros_(self.bias)


class ParallelEmbedding(nn.Embedding):
    def __init__(self, num_embeddings, embedding_dim, init_method=None, **kw):
        super().__init__(num_embeddings, embedding_dim)
        _init(self.weight, init_method)
