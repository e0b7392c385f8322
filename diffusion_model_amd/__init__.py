"""MI355X-native reverse-diffusion sampling: drop-in for the reference's
diffusion.model / diffusion.schedulers / diffusion.sampler classes."""
from .model import UNetMini, Block, SinusoidalPositionEmbeddings
from .schedulers import DDPMScheduler, DDIMScheduler
from .sampler import AncestralSampler, DDIMSampler, create_sampler
from .conditioning import ClassConditioner, create_conditioner
from .utils import get_device, set_seed, count_parameters

__all__ = ["UNetMini", "Block", "SinusoidalPositionEmbeddings", "DDPMScheduler", "DDIMScheduler",
           "ClassConditioner", "create_conditioner", "AncestralSampler", "DDIMSampler", "create_sampler", "get_device", "set_seed",
           "count_parameters"]
