"""build_engine: same entry point as the reference's networks/engines/__init__.py:5-21."""
from .aot_engine import AOTEngine, AOTInferEngine


def build_engine(name, phase='train', **kwargs):
    if name != 'aotengine':
        raise NotImplementedError(f'engine {name!r}: only "aotengine" (LSTT path) is built')
    if phase != 'eval':
        raise NotImplementedError('only the inference engine (phase="eval") is built; training is out of scope')
    return AOTInferEngine(**kwargs)
