"""Task-head FFN named in the north star (reference esme/layer.py:4-23).

`FeedForward` (Linear -> ReLU -> Linear(hidden, 1), fp32 by default) is used only
by the reference's fine-tuning workflows; no LM forward executes it, so it is a
host-side torch module here, not a kernel."""
from torch import nn


class FeedForward(nn.Sequential):
    def __init__(self, embed_dim: int, hidden_dim: int):
        super().__init__()
        self.add_module('linear1', nn.Linear(embed_dim, hidden_dim))
        self.add_module('relu', nn.ReLU())
        self.add_module('linear2', nn.Linear(hidden_dim, 1))
