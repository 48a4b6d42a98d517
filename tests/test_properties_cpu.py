"""Property tests (hypothesis) of the oracle's semantics the parity depends on, and
argument-validation paths of the C-ABI that need no GPU."""
import ctypes

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import restate as R
from racformer_amd import synthetic as syn


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_msmv_out_of_range_points_give_exact_zero_and_weights_are_linear(seed):
    rng = np.random.default_rng(seed)
    S, N, Q, P, C = 2, 3, 3, 4, 4
    feats = [torch.from_numpy(rng.standard_normal((S, N, h, w, C), dtype=np.float32)) for h, w in ((6, 5), (3, 2))]
    loc = torch.from_numpy(rng.random((S, Q, P, 3), dtype=np.float32))
    loc[..., 2] = torch.from_numpy(rng.integers(0, N, (S, Q, P)).astype(np.float32)) / (N - 1)
    w = torch.from_numpy(rng.random((S, Q, P, 2), dtype=np.float32))
    out = R.msmv_gather(feats, loc, w)
    assert torch.allclose(R.msmv_gather(feats, loc, 3 * w), 3 * out, rtol=1e-5, atol=1e-6)
    far = loc.clone()
    far[..., 0] = 1.0 + 2.0 / 1 + float(rng.random())           # more than one pixel outside every level
    assert R.msmv_gather(feats, far, w).abs().max().item() == 0.0
    assert torch.allclose(R.msmv_gather(feats, loc, w, force_torch=True), out, atol=1e-6)


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_view_selection_is_first_valid_view(seed):
    rng = np.random.default_rng(seed)
    T, N, Q, GP = 2, 6, 5, 8
    pts = torch.from_numpy((rng.standard_normal((1, Q, T, GP, 3)) * np.array([25, 25, 1.5])).astype(np.float32))
    l2i = torch.from_numpy(np.asarray(syn.ring_lidar2img(T, N)).astype(np.float32))[None]
    loc, i_view, any_valid = R.project_select(pts, l2i, 256, 704)
    m = l2i.reshape(1, T, N, 4, 4)
    for t in range(T):
        for q in range(Q):
            for p in range(GP):
                x = torch.cat([pts[0, q, t, p], torch.ones(1)])
                valid = []
                for n in range(N):
                    cam = m[0, t, n] @ x
                    hz = max(cam[2].item(), 1e-5)
                    u, v = cam[0].item() / hz / 704, cam[1].item() / hz / 256
                    valid.append(cam[2].item() > 1e-5 and 0 < u < 1 and 0 < v < 1)
                want = valid.index(True) if any(valid) else 0
                # float32 border cases aside (none at these magnitudes), the choice is the first valid view
                assert int(i_view[0, t, q, p]) == want
                assert abs(loc[0, t, q, p, 2].item() - want / (N - 1)) < 1e-6


def test_slot_order_quirk_weights_use_b_g_t_flattening():
    """sampling_4d: slot s=(t*G+g) of the features reads weight slot (g'=s//T, t'=s%T) (Appendix A, Q1)."""
    rng = np.random.default_rng(0)
    B, Q, T, G, P, L, N, C = 1, 2, 3, 4, 2, 2, 3, 4
    feats = [torch.from_numpy(rng.standard_normal((B * T * G, N, h, w, C), dtype=np.float32)) for h, w in ((8, 22), (4, 11))]
    pts = torch.from_numpy((rng.standard_normal((B, Q, T, G, P, 3)) * np.array([15, 15, 0.3]) + np.array([0, 0, 1.0])).astype(np.float32))
    l2i = torch.from_numpy(np.asarray(syn.ring_lidar2img(T, N, (64, 176))).astype(np.float32))[None]
    sw = torch.zeros(B, Q, G, T, P, L)
    gq, tq = 2, 1                                   # switch on exactly one (g', t') weight block
    sw[:, :, gq, tq] = 1.0
    out = R.sampling_4d(pts, feats, sw, l2i, 64, 176).reshape(B, Q, G, T, P, C)
    s_prime = gq * T + tq                           # position in the (b,g,t) flattening
    t_hit, g_hit = s_prime // G, s_prime % G        # the (t,g) feature slot that consumes it
    mask = torch.zeros(G, T, dtype=torch.bool)
    mask[g_hit, t_hit] = True
    assert out[:, :, ~mask].abs().max().item() == 0.0


def test_c_abi_argument_errors_without_gpu():
    from racformer_amd import _lib
    lib = _lib.lib()
    one = (ctypes.c_void_p * 4)(8, 8, 8, 8)
    hw = (ctypes.c_int32 * 8)(4, 4, 4, 4, 4, 4, 4, 4)
    p8 = ctypes.c_void_p(8)
    pc = (ctypes.c_float * 6)(*syn.PC_RANGE)
    db = (ctypes.c_float * 3)(-0.1, 0.0, 0.1)
    rc = lib.rac_sampling4d_fwd(one, hw, 4, p8, p8, p8, p8, p8, p8, p8, p8, None, None, None, 144, 3, 1536, 1, 8, 6, 4, 900, 4, 3, 32,
                                pc, db, 0.08, 256.0, 704.0, 1e-5, 0, -1, None)
    assert rc == -1 and b"64 channels" in lib.rac_last_error()
    rc = lib.rac_sasa_fwd(p8, p8, p8, None, p8, 776, 8, 1, 900, 8, 16, pc, None)
    assert rc == -1 and b"head dim" in lib.rac_last_error()
    rc = lib.rac_mixing_fwd(p8, p8, 1.0, p8, None, 1.0, 65536, 900, 4, 97, 64, 128, 1e-5, 0, None)
    assert rc == -1 and b"in_points" in lib.rac_last_error()
    p16 = ctypes.c_void_p(16)
    rc = lib.rac_bev_sampling_fwd(p8, p8, p16, p8, p8, p8, p8, p8, p8, None, 160, 5, 80, 8, 1, 8, 900, 4, 9, 5, 128, 128, 64,
                                  pc, db, 0.08, 0, None)
    assert rc == -1 and b"staging roles" in lib.rac_last_error()
    rc = lib.rac_bev_sampling_fwd(p8, p8, p8, p8, p8, p8, p8, p8, p8, None, 160, 5, 80, 8, 1, 8, 900, 4, 4, 5, 128, 128, 64,
                                  pc, db, 0.08, 0, None)
    assert rc == -1 and b"16-byte aligned" in lib.rac_last_error()
    rc = lib.rac_add_ln_fwd(p8, 1, 0, 258, 1.0, None, None, p8, p8, None, p8, 258, 4, 258, 1e-5, 0, None, 1.0, 0, 0, None)
    assert rc == -1 and b"dim" in lib.rac_last_error()
    rc = lib.rac_add_ln_fwd(p8, 1, 0, 256, 1.0, None, None, p8, p8, None, p8, 256, 4, 256, 1e-5, 0, p8, 1.0, 4, 1, None)
    assert rc == -1 and b"split_layout" in lib.rac_last_error()
    # entry points added in round 2
    rc = lib.rac_value_proj_fwd(p16, p16, 1.0, None, None, p16, 8, 128, 16384, 256, None)
    assert rc == -1 and b"256 -> 256" in lib.rac_last_error()
    rc = lib.rac_value_proj_fwd(p16, p16, 1.0, None, None, p16, 8, 256, 1000, 256, None)
    assert rc == -1 and b"multiple of 32" in lib.rac_last_error()
    rc = lib.rac_head_finish_fwd(p8, 100, p8, p16, 10, 9, pc, None)
    assert rc == -1 and b"code_size" in lib.rac_last_error()
    rc = lib.rac_conv_pack_bias_fwd(p16, None, p8, p16, 8, 64, 128, 128, 320, 256, 3, 2, None)
    assert rc == -1 and b"groups of 3" in lib.rac_last_error()
    rc = lib.rac_generator_fwd(p16, p16, None, 1.0, p16, 2189, 900, 2189, 256, None)
    assert rc == -1 and b"ld_out" in lib.rac_last_error()
    assert lib.rac_value_proj_fwd(None, None, 1.0, None, None, None, 0, 256, 16384, 256, None) == 0
    # empty problems return success before touching any pointer
    assert lib.rac_msmv_fwd(None, None, 4, None, None, None, 0, 6, 900, 12, 64, 0, 0, 1, 1, None) == 0
    assert lib.rac_bev_pool_v2_fwd(None, None, None, None, None, None, None, None, 64, 0, None) == 0
