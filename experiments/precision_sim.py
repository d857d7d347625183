"""CPU simulation of 16-bit storage policies in the ResNet-50 encoder (the dominant term of the HIP path's error budget,
tests/test_stage_budget.py): which activations may be rounded to bfloat16 / IEEE half, and what does keeping the identity path
of the bottlenecks in fp32 buy?  fp32 arithmetic with the rounding points of a policy inserted; error = rms(d) / std(ref) per stage.
Usage: python experiments/precision_sim.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_ocu_amd.synth import make_clip            # noqa: E402
from rmem_ocu_amd.weights import synth_state_dict   # noqa: E402

R50_BLOCKS, R50_STRIDES = (3, 4, 6), (1, 2, 2)


def fold(w, p_conv, p_bn):
    s = w[p_bn + '.weight'] * (w[p_bn + '.running_var'] + 1e-5).rsqrt()
    return w[p_conv] * s.view(-1, 1, 1, 1), w[p_bn + '.bias'] - w[p_bn + '.running_mean'] * s


def run(img, w, q, qw, ident_q, post_q=None):
    """q: rounding of stored branch activations; qw: of weights; ident_q: of the block outputs on the identity path."""
    cw, cb = fold(w, 'encoder.conv1.weight', 'encoder.bn1')
    x = q(F.relu(F.conv2d(q(img), qw(cw), cb, stride=2, padding=3)))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    xi = x                      # identity-path value (possibly higher precision), x = what convs read
    for li, (nblk, stride) in enumerate(zip(R50_BLOCKS, R50_STRIDES), start=1):
        for b in range(nblk):
            p = f'encoder.layer{li}.{b}'
            s = stride if b == 0 else 1
            w1, b1 = fold(w, p + '.conv1.weight', p + '.bn1')
            w2, b2 = fold(w, p + '.conv2.weight', p + '.bn2')
            w3, b3 = fold(w, p + '.conv3.weight', p + '.bn3')
            o = q(F.relu(F.conv2d(x, qw(w1), b1)))
            o = q(F.relu(F.conv2d(o, qw(w2), b2, stride=s, padding=1)))
            o = F.conv2d(o, qw(w3), b3)
            if p + '.downsample.0.weight' in w:
                wd, bd = fold(w, p + '.downsample.0.weight', p + '.downsample.1')
                idn = F.conv2d(x, qw(wd), bd, stride=s)
            else:
                idn = xi
            y = F.relu(o + idn)
            xi = ident_q(y)
            x = q(y)
        outs.append(x)
    return outs


def main():
    torch.set_num_threads(8)
    w = {k: v.float() for k, v in synth_state_dict(0).items() if k.startswith('encoder.')}
    frames, _ = make_clip(21, 2, 481, 849, 3)
    img = frames[1:2]
    f32 = lambda t: t                                       # noqa: E731
    bf = lambda t: t.to(torch.bfloat16).float()             # noqa: E731
    fp = lambda t: t.to(torch.float16).float()              # noqa: E731
    with torch.no_grad():
        ref = run(img, w, f32, f32, f32)
        for name, q, qw, iq in [('bf16 everywhere', bf, bf, bf), ('fp16 everywhere', fp, fp, fp),
                                ('bf16 branch + fp32 identity', bf, bf, f32), ('fp16 branch + fp32 identity', fp, fp, f32),
                                ('bf16 branch + fp16 identity', bf, bf, fp), ('bf16 act, fp16 weights', bf, fp, bf),
                                ('fp16 act, fp32 weights', fp, f32, fp), ('fp32 act, fp16 weights', f32, fp, f32),
                                ('fp32 act, bf16 weights', f32, bf, f32)]:
            out = run(img, w, q, qw, iq)
            errs = [((o - r).pow(2).mean().sqrt() / r.std()).item() for o, r in zip(out, ref)]
            print(f'{name:32s} stage errors (rms / std): ' + '  '.join(f'{100 * e:.4f} %' for e in errs), flush=True)


if __name__ == '__main__':
    main()
