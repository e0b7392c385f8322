"""get_device / set_seed with the reference's semantics (diffusion/utils.py:78-92, 235-247)."""
import random

import numpy as np
import torch


def get_device():
    """cuda (ROCm shows up as cuda) -> mps -> cpu, as diffusion/utils.py:235-247."""
    if torch.cuda.is_available():
        return torch.device("cuda")
    if hasattr(torch.backends, "mps") and torch.backends.mps.is_available():
        return torch.device("mps")
    return torch.device("cpu")


def set_seed(seed: int):
    """diffusion/utils.py:78-92: seeds python, numpy, torch and torch.cuda generators."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
