"""The arithmetic of the fused network kernel (nuzero_amd/csrc/net_dev.hpp), restated in numpy/torch on the CPU:
every float32 is split exactly into three bf16 pieces, a float32 product is the sum of six piece products, and a
network evaluated that way is as close to the float64-accumulated result as a plain float32 convolution is.
(The GPU side of the same claim is tests/test_gpu_parity.py: kernel vs the reference's outputs within 1e-5.)"""
import numpy as np
import torch
import torch.nn.functional as F


def split3_trunc(x):
    """x = p0 + p1 + p2, each the next 8 significant bits (truncation), as engine.hip split3 / net_dev.hpp split_pair."""
    x = np.asarray(x, np.float32)
    pieces, r = [], x.copy()
    for _ in range(3):
        t = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        pieces.append(t)
        r = r - t
    return pieces, r


def test_three_way_split_is_exact_and_pieces_are_bf16():
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.standard_normal(20000).astype(np.float32) * 10.0 ** rs.randint(-20, 20, 20000),
                        np.float32([0.0, -0.0, 1.0, -1.0, 3.4e38, -3.4e38, 1.17549435e-38, 65504.0, 1e-30])]).astype(np.float32)
    (p0, p1, p2), rest = split3_trunc(x)
    assert np.all(rest == 0.0)                                  # nothing is left after three pieces
    s = (p0.astype(np.float64) + p1.astype(np.float64)) + p2.astype(np.float64)
    assert np.array_equal(s.astype(np.float32), x) and np.array_equal(s, x.astype(np.float64))
    for p in (p0, p1, p2):                                      # each piece has an all-zero low half: a bf16 number
        assert np.all((p.view(np.uint32) & np.uint32(0xFFFF)) == 0)
    # the float32 sum in the epilogue's order reproduces the value bit for bit (residual reads); only the sign of a
    # negative zero is lost ((-0 + 0) + 0 = +0), which no later sum can see
    back = (p0 + p1) + p2
    nz = x != 0.0
    assert np.array_equal(back[nz].view(np.uint32), x[nz].view(np.uint32)) and np.all(back[~nz] == 0.0)


def test_six_term_product_is_within_float32_rounding():
    rs = np.random.RandomState(1)
    a = rs.standard_normal(100000).astype(np.float32)
    w = (rs.standard_normal(100000) * 0.05).astype(np.float32)
    (a0, a1, a2), _ = split3_trunc(a)
    (w0, w1, w2), _ = split3_trunc(w)
    d = np.float64
    six = (a1.astype(d) * w1 + a2.astype(d) * w0 + a0.astype(d) * w2 + a1.astype(d) * w0 + a0.astype(d) * w1 + a0.astype(d) * w0)
    exact = a.astype(d) * w.astype(d)
    rel = np.abs(six - exact) / np.maximum(np.abs(exact), 1e-300)
    assert rel.max() < 2.0 ** -21                               # dropped terms a1w2 + a2w1 + a2w2 < 3 * 2^-23 |aw|
    f32 = (a * w).astype(d)                                     # what one float32 multiplication keeps
    assert np.abs(six - exact).mean() < 4 * np.abs(f32 - exact).mean()


def test_split_network_is_as_accurate_as_a_float32_network():
    from nuzero_amd.weights import synthetic_recurrent_net_weights
    w = {k: torch.from_numpy(v) for k, v in synthetic_recurrent_net_weights(1, 2, 1, 64, 2, True, 3.0).items()}
    rs = np.random.RandomState(0)
    x = torch.from_numpy((rs.random_sample((300, 2, 3, 3)) < 0.35).astype(np.float32))

    def pieces(t):
        (p0, p1, p2), _ = split3_trunc(t.numpy())
        return [torch.from_numpy(p).double() for p in (p0, p1, p2)]

    def conv_split(a, b):
        xa, wb = pieces(a), pieces(b)
        out = 0
        for i, j in ((1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)):
            out = out + F.conv2d(xa[i], wb[j], None, 1, "same")
        return out.float()

    def forward(conv):
        t = F.relu(conv(x, w["projection.0.weight"]))
        for _ in range(2):
            t = conv(torch.cat([t, x], 1), w["recur_module.0.weight"])
            for b in range(2):
                pre = f"recur_module.{1 + b}.before_shortcut."
                t = F.relu(conv(F.relu(conv(t, w[pre + "0.weight"])), w[pre + "2.weight"]) + t)
        p = conv(F.relu(conv(t, w["policy_head.layers.0.weight"])), w["policy_head.layers.2.weight"])
        v = t
        for i in range(4):
            v = conv(v, w[f"value_head.layers.{2 * i}.weight"])
            if i != 3:
                v = torch.tanh(v)
        return torch.softmax(p.reshape(p.shape[0], -1), 1), torch.tanh(v.mean(dim=(1, 2, 3)))

    ref = forward(lambda a, b: F.conv2d(a.double(), b.double(), None, 1, "same").float())
    split = forward(conv_split)
    plain = forward(lambda a, b: F.conv2d(a, b, None, 1, "same"))
    err_split = max(float((split[0] - ref[0]).abs().max()), float((split[1] - ref[1]).abs().max()))
    err_plain = max(float((plain[0] - ref[0]).abs().max()), float((plain[1] - ref[1]).abs().max()))
    assert err_split < 1e-5 and err_split <= 2 * err_plain + 1e-7
