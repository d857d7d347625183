"""build_vos_model: same entry point as the reference's networks/models/__init__.py:5-12."""
from .aot import AOT


def build_vos_model(name, cfg, **kwargs):
    if name == 'aot':
        return AOT(cfg, encoder=cfg.MODEL_ENCODER, **kwargs)
    raise NotImplementedError(f'model {name!r}: only "aot" (the LSTT path BASELINE.json names) is built')
