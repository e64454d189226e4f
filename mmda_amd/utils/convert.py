"""to_gpu / to_cpu of the reference (src/utils/convert.py:4-19) without its per-call argparse."""
import torch

_DEVICE = None


def set_device(device):
    global _DEVICE
    _DEVICE = torch.device(device)


def to_gpu(x, on_cpu=False, gpu_id=None):
    if torch.cuda.is_available() and not on_cpu:
        dev = _DEVICE if _DEVICE is not None else torch.device("cuda")
        x = x.to(dev, non_blocking=True)
    return x


def to_cpu(x):
    if torch.cuda.is_available():
        x = x.to(torch.device("cpu"))
    return x.data
