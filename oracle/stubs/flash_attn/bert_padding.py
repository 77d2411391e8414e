"""Stand-ins for ``flash_attn.bert_padding`` (TEST INFRASTRUCTURE ONLY; see the package docstring): the gather / scatter helpers
the reference's ``_upad_input`` / ``pad_input`` calls use on its half-precision path (lumina_next_t2i/models/model.py:283-336, :404)."""
import torch
import torch.nn.functional as F


def index_first_axis(x, indices):
    return x[indices]


def pad_input(hidden_states, indices, batch, seqlen):
    out = torch.zeros((batch * seqlen,) + tuple(hidden_states.shape[1:]), dtype=hidden_states.dtype)
    out[indices] = hidden_states
    return out.view(batch, seqlen, *hidden_states.shape[1:])


def unpad_input(hidden_states, attention_mask):
    seqlens = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    cu = F.pad(torch.cumsum(seqlens, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape(-1, *hidden_states.shape[2:])
    return flat[indices], indices, cu, int(seqlens.max())
