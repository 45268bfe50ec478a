"""Recursive host->device move of a batch dict (reference: detectron2/utils/memory.py:L13-25)."""
import numpy as np
import torch


def to_cuda(data, device="cuda"):
    if isinstance(data, torch.Tensor):
        return data.to(device, non_blocking=True)
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data).to(device, non_blocking=True)
    if isinstance(data, list):
        return [to_cuda(d, device) for d in data]
    if isinstance(data, tuple):
        return tuple(to_cuda(d, device) for d in data)
    if isinstance(data, dict):
        return {k: to_cuda(v, device) for k, v in data.items()}
    return data


def to_numpy(data):
    if isinstance(data, torch.Tensor):
        return data.detach().cpu().numpy()
    if isinstance(data, list):
        return [to_numpy(d) for d in data]
    if isinstance(data, tuple):
        return tuple(to_numpy(d) for d in data)
    if isinstance(data, dict):
        return {k: to_numpy(v) for k, v in data.items()}
    return data
