"""Fit the non-encoder weights of R50-AOTL on synthetic clips (CPU, oracle as the differentiable model).

No trained checkpoint exists offline, and with purely random weights the 11-way argmax is near-tied almost
everywhere, which makes mask agreement a noisy proxy for numerical parity.  This script takes
``synth_state_dict(0)`` and trains everything except the ResNet-50 encoder (LSTT, decoder, identity bank, temporal
PE, projector) for a few thousand steps on drifting-texture clips whose ground-truth masks are known
(rmem_ocu_amd.synth.clip_masks), so the engine propagates masks confidently -- "trained-like" behaviour.
Output: tests/golden/trained_delta.pt (bf16 tensors of the trained keys only).
    python tests/golden/train_synth_weights.py [steps]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_cpu as O  # noqa: E402
from rmem_ocu_amd.synth import clip_masks, make_clip  # noqa: E402
from rmem_ocu_amd.weights import synth_state_dict  # noqa: E402

H, W = 161, 193
SEQ = 5


def main(steps):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd = synth_state_dict(0)
    train_keys = [k for k in sd if not k.startswith('encoder.') and 'linear_KMem' not in k]
    params = {k: torch.nn.Parameter(sd[k].clone()) for k in train_keys}
    w = dict(sd)
    w.update(params)
    opt = torch.optim.AdamW(params.values(), lr=3e-4, weight_decay=0.0)
    t0 = time.time()
    ema = None
    for step in range(steps):
        seed = 100000 + step
        objs = 1 + step % 4
        stride = 1 + step % 3
        frames, _ = make_clip(seed, SEQ * stride, H, W, objs)
        gts = clip_masks(seed, SEQ * stride, H, W, objs)
        frames, gts = frames[::stride], gts[::stride]
        with torch.no_grad():                      # frozen encoder
            encs = [O.resnet50(frames[i:i + 1], w) for i in range(SEQ)]
        eng = O.OracleEngine(w, 1, 99, 1)
        eng.input_size_2d, eng.enc_size_2d = (H, W), tuple(encs[0][-1].shape[2:])
        eng.pos_emb = O.sine_pos_emb(*eng.enc_size_2d)
        loss, acc = 0.0, 0.0

        def run(i, id_emb, save):
            xs = list(encs[i]) + [F.conv2d(encs[i][-1], w['encoder_projector.weight'], w['encoder_projector.bias'])]
            outs = eng._lstt(xs, id_emb, save)
            return eng._decode(xs, outs, (H, W))

        oh, _ = O.one_hot_mask(gts[0][None, None])
        logit = run(0, O.assign_identity(oh, None, w), False)
        eng.long_mem = [[k, v] for k, v in eng.lstt_long]
        eng.short_mem = [[k, v] for k, v in eng.lstt_short]
        eng.last_mem_step, eng.frame_step = 0, 0
        loss = loss + 0.5 * F.cross_entropy(logit, gts[0][None])
        for i in range(1, SEQ):
            eng.frame_step += 1
            logit = run(i, None, True)
            loss = loss + F.cross_entropy(logit, gts[i][None])
            acc += (logit.argmax(1) == gts[i][None]).float().mean().item() / (SEQ - 1)
            eng.update_memory(gts[i][None, None].float())       # teacher forcing
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params.values(), 5.0)
        opt.step()
        ema = acc if ema is None else 0.95 * ema + 0.05 * acc
        if step % 20 == 0:
            print(f'step {step} loss {loss.item():.3f} acc {acc:.3f} ema {ema:.3f} {time.time() - t0:.0f}s', flush=True)
        if step % 200 == 199 or step == steps - 1:
            torch.save({k: v.detach().to(torch.bfloat16) for k, v in params.items()}, os.path.join(HERE, 'trained_delta.pt'))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
