"""DeAOT engines: networks/engines/deaot_engine.py:9-64 (thin subclasses of the AOT engines on a DeAOT model).

The one behavioural difference sits in the model's memory restriction, not in the engine: DualBranchGPM.restrict_long_memories
has no "bank not full yet" early return (layers/transformer.py:880-892 against 331-333), so the EMA scores and visit counts of
the eviction policy move on EVERY long-term update; an entry is dropped only once the bank overflows.
"""
from __future__ import annotations

from .aot_engine import AOTEngine, AOTInferEngine


class DeAOTEngine(AOTEngine):
    policy_every_update = True

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1, layer_loss_scaling_ratio=2.):
        super().__init__(aot_model, gpu_id, long_term_mem_gap, short_term_mem_skip)
        self.layer_loss_scaling_ratio = layer_loss_scaling_ratio


class DeAOTInferEngine(AOTInferEngine):
    ENGINE = DeAOTEngine
