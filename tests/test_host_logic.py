"""Host-side logic of the engine (no GPU): temporal-PE slots, positional embedding, the eviction
policy against the oracle's restatement, weight packing, the state_dict contract, clip sharding."""
import numpy as np
import torch

from oracle import ref_cpu as O


def test_temporal_slots_match_reference_fixture(golden_ops):
    from rmem_ocu_amd.runtime import temporal_slots
    assert temporal_slots(1) == [0]
    for T in range(2, 33):
        assert temporal_slots(T) == golden_ops[f'slots_T{T}'].tolist(), T


def test_sine_pos_emb_matches_reference_fixture(golden_ops):
    from rmem_ocu_amd.runtime import sine_pos_emb
    assert np.abs(sine_pos_emb(6, 7).numpy() - golden_ops['sine_pos_6x7']).max() < 1e-6
    rows = sine_pos_emb(31, 54)[[0, 53, 54, 800, 1673]].numpy()
    assert np.abs(rows - golden_ops['sine_pos_31x54_rows']).max() < 1e-6


def test_memory_policy_equals_oracle_policy():
    """MemoryPolicy (product) and oracle.choose_eviction walk the same random score streams identically."""
    from rmem_ocu_amd.networks.engines.aot_engine import MemoryPolicy
    rng = np.random.default_rng(0)
    for trial in range(20):
        N = int(rng.integers(2, 9))
        pol, st = MemoryPolicy(), O.EvictionState()
        idx_a, idx_b = list(range(0, 2 * N, 2)), list(range(0, 2 * N, 2))
        frame = 2 * N
        for step in range(30):
            idx_a.append(frame)
            idx_b.append(frame)
            frame += int(rng.integers(1, 4))
            hw = 50
            mass = torch.from_numpy(rng.random((hw, N)).astype(np.float32))
            fg = torch.from_numpy(rng.random(hw).astype(np.float32))
            da = pol.choose((mass * fg[:, None]).sum(0), idx_a)
            db = O.choose_eviction(mass, fg, idx_b, st)
            assert da == db and 1 <= da <= N
            del idx_a[da]
            del idx_b[db]
            assert idx_a == idx_b and idx_a[0] == 0


def test_state_dict_contract_and_packing():
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.pack import pack_state_dict
    from rmem_ocu_amd.weights import synth_state_dict
    cfg = get_config()
    model = build_vos_model('aot', cfg)
    sd = model.state_dict()
    ref = synth_state_dict(0)
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 362
    for k in ref:
        assert sd[k].shape == ref[k].shape
    for attr in ('encoder', 'encoder_projector', 'LSTT', 'decoder', 'patch_wise_id_bank', 'cur_pos_emb', 'mem_pos_emb', 'cfg', 'max_obj_num'):
        assert hasattr(model, attr)
    other = synth_state_dict(7)
    model.load_state_dict(other)
    assert torch.equal(model.state_dict()['LSTT.layers.2.linear_Q.weight'], other['LSTT.layers.2.linear_Q.weight'])
    P = pack_state_dict(ref, torch.device('cpu'))
    assert P['stem.w'].shape == (64, 7, 7, 8) and P['stem.w'][..., 3:].abs().max() == 0
    assert P['idbank.w'].shape == (256, 17, 17, 16) and P['l0.self_qk.w'].shape == (512, 256)
    assert P['l1.dw.w'].shape == (25, 1024) and P['pe_mem'].shape == (4, 256)
    # BN folding: conv(x)*scale + shift == frozen_bn(conv(x))
    x = torch.randn(1, 64, 9, 9)
    p = 'encoder.layer1.0'
    y_ref = O.frozen_bn(torch.nn.functional.conv2d(x, ref[p + '.conv1.weight']), ref, p + '.bn1')
    w = P[p + '.conv1.w'].float().permute(0, 3, 1, 2)
    y = torch.nn.functional.conv2d(x, w, P[p + '.conv1.b'])
    assert (y - y_ref).abs().max() < 0.05 * y_ref.abs().max()
    # conv3 + strided 1x1 shortcut packed side by side (rmem_conv1x1_dual_nhwc): [h | x sampled] @ Wcat^T + b == bn3(conv3(h)) + bn_d(conv_d(x))
    for p, stride in (('encoder.layer1.0', 1), ('encoder.layer2.0', 2)):
        k1, k2 = ref[p + '.conv3.weight'].shape[1], ref[p + '.downsample.0.weight'].shape[1]
        h, xin = torch.randn(1, k1, 5, 6), torch.randn(1, k2, 5 * stride - (stride - 1), 6 * stride - (stride - 1))
        y_ref = O.frozen_bn(torch.nn.functional.conv2d(h, ref[p + '.conv3.weight']), ref, p + '.bn3') + \
            O.frozen_bn(torch.nn.functional.conv2d(xin, ref[p + '.downsample.0.weight'], stride=stride), ref, p + '.downsample.1')
        wcat = P[p + '.c3ds.w'].float()
        assert wcat.shape == (ref[p + '.conv3.weight'].shape[0], k1 + k2)
        a = torch.cat([h, xin[:, :, ::stride, ::stride]], 1).permute(0, 2, 3, 1).reshape(-1, k1 + k2)
        y = (a @ wcat.t() + P[p + '.c3ds.b']).reshape(1, 5, 6, -1).permute(0, 3, 1, 2)
        assert (y - y_ref).abs().max() < 0.02 * y_ref.abs().max()


def test_network_size_rule():
    from rmem_ocu_amd.synth import network_size
    assert network_size(480, 854) == (481, 849)          # cfg 2
    assert network_size(480, 854, scale=1.3) == (625, 1105)   # TEST_MULTISCALE entry (video_transforms.py:604-615)
    assert network_size(720, 1280) == (577, 1041)        # cfg 3
    assert network_size(480, 854, align_corners=False) == (480, 848)


def test_shard_clips_partitions_exactly():
    from rmem_ocu_amd.clip_runner import shard_clips
    lengths = [36, 80, 600, 12, 90, 300, 45, 45, 80, 80, 7]
    for world in (1, 2, 3, 8):
        parts = [shard_clips(len(lengths), r, world, lengths) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lengths)
    assert shard_clips(5, 0, 2) == [0, 2, 4] and shard_clips(5, 1, 2) == [1, 3]


def test_oracle_iou_metric():
    a = np.zeros((4, 4), int); b = np.zeros((4, 4), int)
    a[:2] = 1; b[:3] = 1
    assert abs(O.db_eval_iou(a, b) - 8 / 12) < 1e-9 and O.db_eval_iou(np.zeros((2, 2)), np.zeros((2, 2))) == 1.0


def test_davis_palette_and_png_writer(tmp_path):
    """utils/image.py:8-62 palette head (background black, id 1 maroon, id 2 green, ...) and the indexed PNG writer."""
    from PIL import Image
    from rmem_ocu_amd.evaluator import _davis_palette, save_mask
    pal = _davis_palette()
    assert pal[:15] == [0, 0, 0, 128, 0, 0, 0, 128, 0, 128, 128, 0, 0, 0, 128] and len(pal) == 768
    m = np.zeros((5, 7), np.uint8); m[1:3, 2:5] = 2
    save_mask(m, str(tmp_path / 'a.png'), squeeze_idx=[0, 4, 9])
    back = np.array(Image.open(str(tmp_path / 'a.png')))
    assert back.max() == 9 and (back == 9).sum() == 6


def test_oracle_iou_metric_matches_reference_fixture():
    """f4 pinned: oracle.db_eval_iou against J values the reference's own evaluation/source/metrics.py produced
    (tests/golden/make_golden.py iou) -- several ids per pair, void pixels, an id absent from both maps (J = 1)."""
    import os
    from conftest import GOLDEN
    from oracle import ref_cpu as O
    g = np.load(os.path.join(GOLDEN, 'iou.npz'))
    for i in range(int(g['n'])):
        gt, pred, js, void = g[f'gt{i}'], g[f'pred{i}'], g[f'j{i}'], g[f'void{i}']
        v = void if void.size else None
        for k, j in enumerate(js, start=1):
            assert abs(O.db_eval_iou(gt == k, pred == k, v) - j) < 1e-12, (i, k)


def test_pack_frag_layout():
    """pack.pack_frag: element (n, k) of an [N, K] weight sits where include/rmem.h says the chain kernels read it."""
    import torch
    from rmem_ocu_amd.pack import pack_frag
    N, K = 512, 64
    w = (torch.arange(N)[:, None] * 1000 + torch.arange(K)[None, :]).float()
    for nw in (4, 8):
        f = pack_frag(w, nw).reshape(N // 256, nw, K // 32, 16 // nw, 64, 8)
        for nb, wave, kc, j, lane, e in [(0, 0, 0, 0, 0, 0), (1, 3, 1, 1, 37, 5), (0, 2, 1, 16 // nw - 1, 63, 7), (1, nw - 1, 0, 1, 16, 0)]:
            n = 256 * nb + (256 // nw) * wave + 16 * j + (lane & 15)
            k = 32 * kc + 8 * (lane >> 4) + e
            assert f[nb, wave, kc, j, lane, e].item() == n * 1000 + k


def test_deaot_group_key_table_rows_per_frame():
    """Rows per memory frame of a DeAOT group's key table (group_runtime_deaot.rows_per_frame): one row per frame once the clips fill
    the GPU, never more than 4 per frame or 32 in all (the table rmem_gated_attn records mass for), at least 1."""
    from rmem_ocu_amd.group_runtime_deaot import rows_per_frame
    L = 31 * 54
    assert rows_per_frame(L, 9, 8) == 1            # the bench shape: 14 query tiles x 9 frames x 8 clips = 1008 workgroups
    assert rows_per_frame(L, 9, 1) == 3            # single clip: 32 // 9
    assert rows_per_frame(L, 1, 8) == 4            # first frames of a clip: few keys, cut as far as allowed
    assert rows_per_frame(L, 4, 8) == 2
    for T in range(1, 33):
        for clips in (1, 2, 4, 8, 16):
            r = rows_per_frame(L, T, clips)
            assert 1 <= r <= 4 and r * T <= 32
            if r > 1:                                # cut only while the launch is short of ~512 workgroups
                assert ((L + 127) // 128) * T * clips * (r - 1) < 512
