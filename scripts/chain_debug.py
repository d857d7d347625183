"""Debug: which buffers differ between the chained and the unfused LSTT launch lists (per key, both flavours)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmem_ocu_amd import build_vos_model, get_config, ops
from rmem_ocu_amd.group_runtime import GroupRuntime
from rmem_ocu_amd.weights import synth_state_dict

dev = torch.device('cuda', 0)
for dt in sys.argv[1:] or ['bf16', 'fp16']:
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.MODEL_DTYPE = dt
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    P = model.packed()
    B, T = 3, 2
    outs = []
    for chain in (False, True, True):
        rt = GroupRuntime(P, (161, 193), 4, dev, B, lookahead=1)
        rt.chain = chain
        rt.chain_stats = False
        L = rt.L
        g = torch.Generator().manual_seed(7)
        r = lambda *s: torch.randn(*s, generator=g)
        rt.x.copy_(r(B * L, 256))
        for i in range(3):
            rt.short_K[i].copy_(r(B * L, 256)); rt.short_V[i].copy_(r(B * L, 256))
            rt.bank_K[i].copy_(r(*rt.bank_K[i].shape)); rt.bank_V[i].copy_(r(*rt.bank_V[i].shape))
        rt.slots = [[1, 3] for _ in range(B)]
        s = torch.cuda.current_stream().cuda_stream
        rt.prepare_pos(s)
        rt.upload_chunks(s)
        nl = int(os.environ.get('NLAUNCH', 0))
        prog = rt.prog_lstt(False, T, True)
        ops.run(prog if not nl else prog[:nl], s)
        torch.cuda.synchronize()
        outs.append({'qkv': rt.qkv.clone(), **{f'cv{i}': rt.curr_V[i].clone() for i in range(3)}, **{f'cq{i}': rt.curr_Q[i].clone() for i in range(3)},
                     'k4': rt.k4.clone(), 'v4': rt.v4.clone(), 'att': rt.att.clone(), 'att2': rt.att2.clone(),
                     **{f'tgt3_{i}': rt.tgt3[i].clone() for i in range(3)}, 'h1': rt.h1.clone(), 'h3': rt.h3.clone(), 'x': rt.x.clone(),
                     'dec_in': rt.dec_in[:, 256:].clone()})
    for k in outs[0]:
        a, b, c = outs[0][k], outs[1][k], outs[2][k]
        print(f'{dt} {k:8s} unfused-vs-chain differ {int((a != b).sum()):8d} / {a.numel()}  max|d| {(a.float() - b.float()).abs().max().item():.3e}   chain-vs-chain differ {int((b != c).sum())}'
              f'   finite {bool(torch.isfinite(a.float()).all())} {bool(torch.isfinite(b.float()).all())}')
