"""Configuration bag with the reference's field names (reference src/config.py:71-170).

``get_config`` mirrors the reference's argparse flags for the hot path; dataset paths, wandb names and the BERT
switch are out of scope.  ``optimizer``/``activation`` are looked up by name like the reference
does (config.py:24-27), but resolve to this package's HIP-backed classes / activation ids.
"""
from __future__ import annotations

import argparse
from datetime import datetime

ACTIVATIONS = ("elu", "hardshrink", "hardtanh", "leakyrelu", "prelu", "relu", "rrelu", "tanh")      # reference config.py:25-27


def str2bool(v):
    """reference config.py:61-68"""
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


class Config(object):
    """Attribute bag (reference config.py:71-96).  'optimizer' names resolve to mmda_amd.optim classes."""

    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            if key == "optimizer" and isinstance(value, str):
                from . import optim
                value = optim.optimizer_dict[value]
            if key == "activation":
                value = activation_name(value)
            setattr(self, key, value)

    def __str__(self):
        import pprint
        return "Configurations\n" + pprint.pformat(self.__dict__)


def activation_name(act) -> str:
    """Accepts the reference's forms: a name ('leakyrelu'), an nn.Module class (nn.LeakyReLU, config.py:25-27) or
    an instance; returns the lower-case name."""
    if isinstance(act, str):
        name = act.lower()
    else:
        cls = act if isinstance(act, type) else type(act)
        name = cls.__name__.lower()
    if name not in ACTIVATIONS:
        raise ValueError(f"unknown activation '{name}'")
    return name


DEFAULTS = dict(
    mode="train", runs=5, use_confidNet=False, device="cuda", eval_mode="macro",
    use_bert=False, use_cmd_sim=True, data="mosei", name="run",
    num_classes=6, batch_size=64, eval_batch_size=10, n_epoch=40, patience=6,
    diff_weight=0.3, sim_weight=0.7, sp_weight=0.0, recon_weight=0.7, conf_weight=0.3,
    learning_rate=1e-4, optimizer="Adam", clip=1.0, weight_decay=0.1,
    extractor="lstm", rnncell="lstm", embedding_size=300, hidden_size=128, dropout=0.1,
    reverse_grad_weight=1.0, activation="leakyrelu", threshold=0.35, model="MISA",
    # added for the MI355X build (not reference flags)
    visual_size=35, acoustic_size=74, vocab_size=20000, precision="bf16", seq_len=50, pretrained_emb=None,
    fusion_fp8=False, dp_global_stats=False,
)


def get_config(parse=True, **optional_kwargs):
    """reference config.py:99-170 (same flags, same defaults except use_bert which is fixed False: the BERT branch
    needs a hub download and is out of scope)."""
    parser = argparse.ArgumentParser()
    for k, v in DEFAULTS.items():
        if isinstance(v, bool):
            parser.add_argument(f"--{k}", type=str2bool, default=v)
        elif v is None:
            parser.add_argument(f"--{k}", default=None)
        else:
            parser.add_argument(f"--{k}", type=type(v), default=v)
    if parse:
        kwargs = vars(parser.parse_args())
    else:
        kwargs = vars(parser.parse_known_args()[0])
    if kwargs.get("name") == "run":
        kwargs["name"] = datetime.now().strftime("%Y-%m-%d_%H:%M:%S")
    kwargs.update(optional_kwargs)
    if kwargs.get("use_bert"):
        raise NotImplementedError("use_bert=True needs bert-base-uncased from the hub; only the GloVe/LSTM text branch is built")
    if "word2id" not in kwargs:
        kwargs["word2id"] = range(kwargs["vocab_size"])        # only len() is read (reference models.py:47)
    return Config(**kwargs)


def make_config(**kw):
    """Programmatic construction with the reference's defaults."""
    d = dict(DEFAULTS)
    d.update(kw)
    if "word2id" not in d:
        d["word2id"] = range(d["vocab_size"])
    return Config(**d)
