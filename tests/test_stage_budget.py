"""Per-stage error budget of the HIP path against the fp32 oracle on the fitted ("trained-like") clips.

Where does the logit difference between the bf16 HIP pipeline and the fp32 reference math come from?  Both engines run a
golden clip teacher-forced (the reference's masks are fed back, so frame i is the same computation on both sides); at the
last frame every stage output of the HIP runtime is compared with the oracle's

    cumulative : what the stage outputs in the real pipeline (its inputs carry the upstream error), and
    intrinsic  : what the stage outputs when its inputs are REPLACED by the oracle's values (rounded to the stage's input
                 type): the error the stage adds by itself -- encoder (same image by construction), LSTT stack (oracle's
                 projector output + the oracle's long / short-term memory written into the bank), FPN decoder (oracle's
                 concat input and encoder shortcuts).

Errors are max |d| and rms(d), both divided by the reference tensor's std.  The asserted bounds are ~1.5x the values
measured on MI355X (DESIGN.md §5 lists them); a regression in one stage shows up in that stage's row, not only as a
slightly worse end-to-end logit.  The table is also written to gpurun_out/stage_budget_<clip>.json."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _rel(got: torch.Tensor, ref: torch.Tensor):
    d = (got.float().cpu() - ref.float().cpu())
    s = ref.float().std().item() + 1e-12
    return d.abs().max().item() / s, d.pow(2).mean().sqrt().item() / s


def _nhwc(t: torch.Tensor) -> torch.Tensor:
    """oracle [1, C, H, W] -> [H*W, C]"""
    return t[0].permute(1, 2, 0).reshape(-1, t.shape[1]).contiguous()


def _committed_budget(name: str, dtype: str):
    """The newest committed MI355X measurement of this clip / flavour (profiles/rNN/stage_budget_<clip>_<dtype>.json, written by this
    test itself into gpurun_out/ and copied from there): every stage's cumulative and intrinsic error is held to 1.5x its rms
    and 2x its maximum (the maximum of a few million elements moves more from run to run than the rms)."""
    prof = os.path.join(ROOT, 'profiles')
    for rnd in sorted((d for d in os.listdir(prof) if d.startswith('r')), reverse=True):
        f = os.path.join(prof, rnd, f'stage_budget_{name[:-4]}_{dtype}.json')
        if os.path.exists(f):
            with open(f) as fh:
                return json.load(fh)['stages'], f
    return None, None


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('name', ['clip_small_fitted.npz', 'clip_full_fitted.npz'])
def test_stage_error_budget(name, dtype):
    if not os.path.exists(os.path.join(GOLDEN, 'trained_delta.pt')):
        pytest.skip('fitted weights missing')
    from oracle import ref_cpu as O
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.weights import fitted_state_dict
    from test_hip_engine import _engine, _load
    g, frames, mask, (former, latter, n, h, w, oh, ow, gap, objs) = _load(name)
    n_run = n if 'small' in name else min(n, 20)        # the full-size oracle costs ~1 s per frame
    dev = torch.device('cuda', 0)
    eng = _engine(former, latter, gap, fitted=True, dtype=dtype)
    e16 = torch.bfloat16 if dtype == 'bf16' else torch.float16
    sd = fitted_state_dict(0)

    class Rec(O.OracleEngine):
        def _lstt(self, xs, id_emb, save_attn):
            self.rec_xs = xs
            self.rec_long = None if self.long_mem is None else [[m.clone() for m in lm] for lm in self.long_mem]
            self.rec_short = None if self.short_mem is None else [[m.clone() for m in sm] for sm in self.short_mem]
            outs = super()._lstt(xs, id_emb, save_attn)
            self.rec_outs = outs
            return outs

    ora = Rec(sd, former, latter, gap)
    fd = frames.to(dev)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        eng.add_reference_frame(fd[0:1], mask.to(dev), obj_nums=[int(mask.max())], frame_step=0)
        ora.add_reference_frame(frames[0:1], mask, 0)
        for i in range(1, n_run):
            logit = eng.match_propogate_one_frame(fd[i:i + 1], output_size=(oh, ow))
            ref_logit = ora.match_propogate_one_frame(frames[i:i + 1], (oh, ow))
            if i == n_run - 1:
                break
            fed = torch.from_numpy(g['labels'][i - 1].astype(np.float32))[None, None]
            m = F.interpolate(fed, size=(h, w), mode='nearest')
            eng.update_memory(m.to(dev))
            ora.update_memory(m)
    torch.cuda.synchronize()
    inner = eng.aot_engines[0]
    rt = inner.rt
    assert inner.long_memories_indexes == ora.long_memories_indexes
    L, C = rt.L, 256
    xs = ora.rec_xs                                      # [enc1, enc2, enc3, projector out]
    ref_dec_in = torch.cat([_nhwc(xs[-1])] + [o.reshape(L, C) for o in ora.rec_outs], dim=1)      # [L, 1024]
    ref_logits4 = _nhwc(ora.pred_id_logits)              # [M4, 11]
    table = {}

    def row(stage, kind, got, ref):
        mx, rms = _rel(got, ref)
        table.setdefault(stage, {})[kind] = {'max': round(mx, 5), 'rms': round(rms, 5)}

    # ---- cumulative: the real pipeline's buffers at this frame ----
    for k, (got, ref) in enumerate(zip((rt.enc1, rt.enc2, rt.enc3), xs[:3])):
        row(f'encoder stage {k + 1}', 'cumulative', got.reshape(-1, ref.shape[1]), _nhwc(ref))
        row(f'encoder stage {k + 1}', 'intrinsic', got.reshape(-1, ref.shape[1]), _nhwc(ref))        # same image on both sides
    di = rt.dec_in.view(L, 4 * C)
    row('projector', 'cumulative', di[:, :C], ref_dec_in[:, :C])
    for i in range(3):
        row(f'LSTT layer {i} (decoder norm)', 'cumulative', di[:, (i + 1) * C:(i + 2) * C], ref_dec_in[:, (i + 1) * C:(i + 2) * C])
    row('logits (1/4 res)', 'cumulative', rt.logits.view(-1, 16)[:, :11], ref_logits4)
    row('logits (output size)', 'cumulative', logit, ref_logit)
    lab, ref_lab = logit.argmax(1).cpu(), ref_logit.argmax(1)
    flips_cum = (lab != ref_lab).float().mean().item()

    s = inner.stream.cuda_stream
    # ---- intrinsic, projector: oracle's encoder stage 3 in, projector out ----
    enc3_save = rt.enc3.clone()
    rt.enc3.copy_(_nhwc(xs[2]).to(dev).to(e16).view_as(rt.enc3))
    ops.run([rt._proj_op(rt.enc_ch[2])], s)
    inner.stream.synchronize()
    row('projector', 'intrinsic', rt.dec_in.view(L, 4 * C)[:, :C], ref_dec_in[:, :C])
    # ---- intrinsic, LSTT stack: oracle's projector output + oracle's memories ----
    rt.x.copy_(_nhwc(xs[-1]).to(dev))
    T = len(rt.slots)
    for i in range(3):
        lk, lv = ora.rec_long[i]
        for t, slot in enumerate(rt.slots):
            rt.bank_K[i][slot].copy_(lk[t].reshape(L, C).to(dev).to(e16))
            rt.bank_V[i][slot].copy_(lv[t].reshape(L, C).to(dev).to(e16))
        sk, sv = ora.rec_short[i]
        rt.short_K[i].copy_(sk.reshape(L, C).to(dev).to(e16))
        rt.short_V[i].copy_(sv.reshape(L, C).to(dev).to(e16))
    torch.cuda.synchronize()
    ops.run(rt.prog_lstt(False, T, want_mass=False), s)
    inner.stream.synchronize()
    for i in range(3):
        row(f'LSTT layer {i} (decoder norm)', 'intrinsic', rt.dec_in.view(L, 4 * C)[:, (i + 1) * C:(i + 2) * C],
            ref_dec_in[:, (i + 1) * C:(i + 2) * C])
    # ---- intrinsic, decoder: oracle's concat input and encoder shortcuts ----
    rt.dec_in.copy_(ref_dec_in.to(dev).to(e16).view_as(rt.dec_in))
    for buf, ref in zip((rt.enc1, rt.enc2, rt.enc3), xs[:3]):
        buf.copy_(_nhwc(ref).to(dev).to(e16).view_as(buf))
    torch.cuda.synchronize()
    ops.run(rt.prog_decode(), s)
    inner.stream.synchronize()
    row('logits (1/4 res)', 'intrinsic', rt.logits.view(-1, 16)[:, :11], ref_logits4)
    rt.enc3.copy_(enc3_save)

    print(f'\n{name} [{dtype}]: frame {n_run - 1}, bank T = {T}, logit std {ref_logits4.std().item():.3f}, label flips {100 * flips_cum:.4f} %')
    print(f'{"stage":34s} {"cumulative max / rms":>24s} {"intrinsic max / rms":>24s}')
    for st, r in table.items():
        c, i = r.get('cumulative'), r.get('intrinsic')
        f = lambda v: '-' if v is None else f'{v["max"]:.4f} / {v["rms"]:.5f}'   # noqa: E731
        print(f'{st:34s} {f(c):>24s} {f(i):>24s}')
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f'stage_budget_{name[:-4]}_{dtype}.json'), 'w') as fjs:
        json.dump({'clip': name, 'dtype': dtype, 'frame': n_run - 1, 'T': T, 'logit_std': ref_logits4.std().item(), 'label_flips': flips_cum,
                   'stages': table}, fjs, indent=1)
    committed, src = _committed_budget(name, dtype)
    assert committed is not None, f'no committed stage budget for {name} [{dtype}] under profiles/'
    for st, r in committed.items():
        for kind, ref in r.items():
            got = table[st][kind]
            assert got['rms'] <= 1.5 * ref['rms'] + 1e-5 and got['max'] <= 2.0 * ref['max'] + 1e-4, (st, kind, got, ref, src)
    # every stage must be at bf16 level: nothing may add more than 10 % rms of its own output scale
    for st, r in table.items():
        for kind, v in r.items():
            assert v['rms'] < 0.1, (st, kind, v)
