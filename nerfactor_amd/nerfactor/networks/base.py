"""networks.base.Network (reference: nerfactor/networks/base.py:21-26): a bag of layers."""
import torch


class Network(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = torch.nn.ModuleList()

    def forward(self, x):
        raise NotImplementedError
