"""build_vos_model: same entry point as the reference's networks/models/__init__.py:5-12."""
from .aot import AOT
from .deaot import DeAOT


def build_vos_model(name, cfg, **kwargs):
    if name == 'aot':
        return AOT(cfg, encoder=cfg.MODEL_ENCODER, **kwargs)
    if name == 'deaot':
        return DeAOT(cfg, encoder=cfg.MODEL_ENCODER, **kwargs)
    raise NotImplementedError(f'model {name!r}: built models are "aot" (LSTT path) and "deaot" (gated propagation path)')
