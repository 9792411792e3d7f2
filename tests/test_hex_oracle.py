"""The hexagonal convolution of oracle/net.py (PARITY UNPINNED: hexagdly is not installed, so it restates the
package's documented addressing) against a plain loop over the neighbourhood SCS_Game itself uses
(oracle/scs.py neighbours, pinned to the reference by tests/test_scs_oracle.py): the seven taps are the cell,
its n / s neighbours (kernel0) and its nw / sw / ne / se neighbours (kernel1 [upper, lower] x [left, right])."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("config", ["mirrored_5x5.yml", "two_types_6x5.yml", "ten_by_ten.yml"])
def test_hex_conv_follows_the_games_adjacency(config):
    from oracle.net import hex_conv2d
    from oracle.scs import ScsConfig, ScsGame
    game = ScsGame(ScsConfig(os.path.join(GOLDEN, "scs_configs", config)))
    rows, cols = game.cfg.rows, game.cfg.cols
    rs = np.random.RandomState(3)
    x = rs.standard_normal((2, 3, rows, cols)).astype(np.float32)
    k0 = rs.standard_normal((4, 3, 3, 1)).astype(np.float32)
    k1 = rs.standard_normal((4, 3, 2, 2)).astype(np.float32)
    got = hex_conv2d(x, k0, k1).numpy()
    want = np.zeros((2, 4, rows, cols), np.float64)
    for r in range(rows):
        for c in range(cols):
            n, ne, se, s, sw, nw = game.neighbours((r, c))
            taps = [((r, c), k0[:, :, 1, 0]), (n, k0[:, :, 0, 0]), (s, k0[:, :, 2, 0]), (nw, k1[:, :, 0, 0]),
                    (sw, k1[:, :, 1, 0]), (ne, k1[:, :, 0, 1]), (se, k1[:, :, 1, 1])]
            for pos, w in taps:
                if pos is not None:
                    want[:, :, r, c] += x[:, :, pos[0], pos[1]].astype(np.float64) @ w.T.astype(np.float64)
    assert np.max(np.abs(got - want)) < 1e-5
