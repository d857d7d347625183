"""rmem_ocu_amd: MI355X-native space-time memory-reading VOS inference engine.

Drop-in surface (mirrors the reference's aot_plus/networks API):
    from rmem_ocu_amd import get_config, build_vos_model, build_engine
The compute path is librmem_hip.so (hand-written HIP for gfx950) -- see include/rmem.h.
"""
from .config import EngineConfig, get_config  # noqa: F401


def build_vos_model(name, cfg, **kw):
    from .networks.models import build_vos_model as f
    return f(name, cfg, **kw)


def build_engine(name, phase='train', **kw):
    from .networks.engines import build_engine as f
    return f(name, phase=phase, **kw)
