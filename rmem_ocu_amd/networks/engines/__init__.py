"""build_engine: same entry point as the reference's networks/engines/__init__.py:5-21."""
from .aot_engine import AOTEngine, AOTInferEngine
from .deaot_engine import DeAOTEngine, DeAOTInferEngine


def build_engine(name, phase='train', **kwargs):
    if name not in ('aotengine', 'deaotengine'):
        raise NotImplementedError(f'engine {name!r}: built engines are "aotengine" and "deaotengine"')
    if phase != 'eval':
        raise NotImplementedError('only the inference engine (phase="eval") is built; training is out of scope')
    return (AOTInferEngine if name == 'aotengine' else DeAOTInferEngine)(**kwargs)
