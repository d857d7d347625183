#!/usr/bin/env python3
"""rmem_lstt_chain_b alone at several (rows per clip, clips, element type): --reps launches back to back per event pair.
Usage: python scripts/chain_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rmem_ocu_amd import ops, pack
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    for dt in (torch.bfloat16, torch.float16):
        for (L, clips) in ((1674, 8), (2442, 8), (3600, 8), (3600, 4), (3600, 2), (1800, 8)):
            R = L * clips
            r = lambda *s, sc=0.5: (torch.randn(*s, generator=g) * sc).to(dt).to(dev)   # noqa: E731
            f = lambda *s: torch.randn(*s, generator=g).to(dev)                          # noqa: E731
            pk = lambda w: pack.pack_frag(w.cpu()).to(dev)                               # noqa: E731
            op = ops.lstt_chain_b(L=L, clips=clips, att_long=r(R, 256), att_short=r(R, 256), x=f(R, 256), w_long=pk(r(256, 256, sc=0.05)),
                                  b_long=f(256), w_short=pk(r(256, 256, sc=0.05)), b_short=f(256), tgt3=r(R, 256), ln3=(f(256), f(256)),
                                  w1=pk(r(1024, 256, sc=0.05)), b1=f(1024), h1=r(R, 1024))
            for _ in range(3):
                ops.run(op)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    ops.run(op)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 8)
            ts.sort()
            print(f'{str(dt):16s} L = {L:5d} clips = {clips}: rows {R:6d}  {ts[2]:8.1f} us', flush=True)


if __name__ == '__main__':
    main()
