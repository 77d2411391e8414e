from . import _absent

index_first_axis = _absent
pad_input = _absent
unpad_input = _absent
