"""Small helpers of the transport package (API mirror of ``lumina_next_t2i/transport/utils.py``)."""
import torch as th


class EasyDict:
    """attribute / item view over a plain dict"""

    def __init__(self, sub_dict):
        self.__dict__.update(sub_dict)

    def __getitem__(self, key):
        return getattr(self, key)


def mean_flat(x):
    """mean over every non-batch axis"""
    return x.flatten(1).mean(dim=1) if x.dim() > 1 else x


def log_state(state):
    lines = []
    for key in sorted(state):
        value = state[key]
        text = str(value)
        lines.append(f"{key}: [{value.__class__.__name__}]" if ("object at" in text or "<object" in text) else f"{key}: {value}")
    return "\n".join(lines)
