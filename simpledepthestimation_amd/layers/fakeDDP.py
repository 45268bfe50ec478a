"""Single-process stand-in for DistributedDataParallel: gives the reference's training scripts the ``model.module`` attribute they reach
through (``model.module.depth_net.encoder`` ...) when no process group exists.  Same contract as detectron2/layers/fakeDDP.py:L4-10."""
from torch import nn


class FakeDDP(nn.Module):
    """Wraps `wrapped` under the attribute name DDP uses; calls, ``train()``/``eval()`` and ``state_dict`` prefixes behave like DDP's."""

    def __init__(self, wrapped: nn.Module):
        nn.Module.__init__(self)
        self.add_module("module", wrapped)

    def forward(self, *batch, **kw):
        return self._modules["module"](*batch, **kw)
