"""``create_transport`` factory - API mirror of ``lumina_next_t2i/transport/__init__.py:4-66``."""
from .transport import ModelType, PathType, Sampler, Transport, WeightType

_PREDICTION = {"noise": ModelType.NOISE, "score": ModelType.SCORE}
_LOSS = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD}
_PATH = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None,
                     snr_type="uniform"):
    """Build the Transport state.  Model prediction defaults to velocity; eps defaults follow the reference:
    VP -> (1e-5, 1e-3); GVP/Linear with a non-velocity model -> (1e-3, 1e-3); velocity on GVP/Linear -> (0, 0)
    (note the reference derives BOTH defaults from ``train_eps is None`` - kept)."""
    model_type = _PREDICTION.get(prediction, ModelType.VELOCITY)
    loss_type = _LOSS.get(loss_weight, WeightType.NONE)
    ptype = _PATH[path_type]
    unset = train_eps is None
    if ptype is PathType.VP:
        train_eps, sample_eps = (1e-5 if unset else train_eps), (1e-3 if unset else sample_eps)
    elif model_type is not ModelType.VELOCITY:
        train_eps, sample_eps = (1e-3 if unset else train_eps), (1e-3 if unset else sample_eps)
    else:
        train_eps = sample_eps = 0
    return Transport(model_type=model_type, path_type=ptype, loss_type=loss_type, train_eps=train_eps,
                     sample_eps=sample_eps, snr_type=snr_type)
