"""``create_transport`` factory - API mirror of ``lumina_next_t2i/transport/__init__.py:4-66``."""
from .transport import ModelType, PathType, Sampler, Transport, WeightType

_PREDICTION = {"noise": ModelType.NOISE, "score": ModelType.SCORE}
_LOSS = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD}
_PATH = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None,
                     snr_type="uniform"):
    """Build the Transport state.  Model prediction defaults to velocity; eps defaults follow the reference:
    VP -> (1e-5, 1e-3); GVP/Linear with a non-velocity model -> (1e-3, 1e-3); velocity on GVP/Linear -> (0, 0)
    The reference tests ``train_eps is None`` a second time AFTER assigning train_eps (__init__.py:47-57), which leaves
    sample_eps = None whenever only the defaults are used on VP / non-velocity models, and its check_interval then fails on
    ``1 - None``; here each eps takes its default when IT is unset (what the reference evidently intends), and an explicit
    sample_eps is kept as given - like the reference, whose second test sees the just-assigned train_eps and leaves
    sample_eps alone.  With explicit eps values the behaviour is identical (tests/golden/transport_kat.npz)."""
    model_type = _PREDICTION.get(prediction, ModelType.VELOCITY)
    loss_type = _LOSS.get(loss_weight, WeightType.NONE)
    ptype = _PATH[path_type]
    if ptype is PathType.VP:
        train_eps, sample_eps = (1e-5 if train_eps is None else train_eps), (1e-3 if sample_eps is None else sample_eps)
    elif model_type is not ModelType.VELOCITY:
        train_eps, sample_eps = (1e-3 if train_eps is None else train_eps), (1e-3 if sample_eps is None else sample_eps)
    else:
        train_eps = sample_eps = 0
    return Transport(model_type=model_type, path_type=ptype, loss_type=loss_type, train_eps=train_eps,
                     sample_eps=sample_eps, snr_type=snr_type)
