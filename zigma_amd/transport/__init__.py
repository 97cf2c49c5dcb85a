"""Flow-matching transport package (mirror of the reference's `transport/`)."""
from .transport import ModelType, PathType, Sampler, Transport, WeightType


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None):
    """Same defaults and epsilon rules as the reference's create_transport (transport/__init__.py:4-75)."""
    try:
        model_type = {"noise": ModelType.NOISE, "score": ModelType.SCORE, "velocity": ModelType.VELOCITY}[prediction]
    except KeyError:
        raise ValueError(f"Model type {prediction} not implemented")
    try:
        loss_type = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD, None: WeightType.NONE}[loss_weight]
    except KeyError:
        raise ValueError(f"Loss type {loss_weight} not implemented")
    path_type = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}[path_type]
    # NB the reference tests `train_eps is None` AFTER reassigning train_eps (transport/__init__.py:51-58), which
    # leaves sample_eps = None for these two branches; the defaults below are the evident intent.
    if path_type in [PathType.VP]:
        train_eps, sample_eps = (1e-5 if train_eps is None else train_eps), (1e-3 if sample_eps is None else sample_eps)
    elif path_type in [PathType.GVP, PathType.LINEAR] and model_type != ModelType.VELOCITY:
        train_eps, sample_eps = (1e-3 if train_eps is None else train_eps), (1e-3 if sample_eps is None else sample_eps)
    else:                                   # velocity & [GVP, LINEAR] is stable everywhere
        train_eps = 0
        sample_eps = 0
    return Transport(model_type=model_type, path_type=path_type, loss_type=loss_type, train_eps=train_eps,
                     sample_eps=sample_eps)
