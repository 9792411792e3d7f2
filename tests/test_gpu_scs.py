"""SCS rules on the device (nz_scs_*: step, legal mask, state image) against the oracle
(oracle/scs.py, pinned to the reference by tests/test_scs_oracle.py), step by step on random
games.  Needs a GPU."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,n_games", [("mirrored_5x5", 48), ("late_reinforcements_5x5", 48),
                                          ("two_types_6x5", 48), ("ten_by_ten", 12)])
def test_scs_device_rules_equal_oracle(name, n_games):
    from nuzero_amd.scs import ScsBatch, ScsGameConfig
    from oracle.scs import ScsConfig, ScsGame
    path = os.path.join(GOLDEN, "scs_configs", name + ".yml")
    ocfg = ScsConfig(path)
    cfg = ScsGameConfig(path)
    assert (cfg.planes, cfg.channels, cfg.num_actions) == (ocfg.planes, ocfg.channels, ocfg.num_actions)
    batch = ScsBatch(cfg, n_games)
    games = [ScsGame(ocfg) for _ in range(n_games)]
    rs = np.random.RandomState(7)
    steps = 0
    while True:
        st = batch.status().cpu().numpy()
        mask = batch.legal_mask().cpu().numpy()
        img = batch.state_image().cpu().numpy()
        actions = np.full(n_games, -1, np.int32)
        for g, og in enumerate(games):
            want = (og.player, og.sub_phase, og.stage, og.turn, int(og.terminal), og.terminal_value, og.length)
            assert tuple(st[g]) == want, (steps, g, tuple(st[g]), want)
            assert np.array_equal(img[g], og.state_image()[0]), (steps, g)
            if og.terminal:
                assert not mask[g].any()
                continue
            om = og.possible_actions()
            assert np.array_equal(mask[g], om), (steps, g)
            actions[g] = rs.choice(np.nonzero(om.reshape(-1))[0])
            og.step_index(int(actions[g]))
        if (actions < 0).all():
            break
        batch.step(actions)
        steps += 1
    assert steps > 20
    outcomes = [g.terminal_value for g in games]
    assert len(outcomes) == n_games
    batch.close()
