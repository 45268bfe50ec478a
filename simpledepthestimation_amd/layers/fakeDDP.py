"""`.module` shim for single-GPU runs (reference: detectron2/layers/fakeDDP.py:L4-10)."""
import torch.nn as nn


class FakeDDP(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.module = model

    def forward(self, x):
        return self.module(x)
