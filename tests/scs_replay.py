"""Test infrastructure: replay SCS self-play games on the CPU oracle with the DEVICE network's own outputs.

`LeafRecorder` wraps an evaluator for ScsSelfPlay.play and keeps, per game, the sequence of leaf evaluations the
device search consumed (one simulation is in flight per tree, Explorer.py:49-61, so the oracle's evaluate() calls of
game g come in exactly that order).  `replay_game` plays the game again with oracle/search.py + oracle/scs.py, feeding
evaluate() the recorded (probs, value) after checking that the oracle's leaf image is the recorded one, and returns the
per-move trace: every root statistic must then equal the device's bit for bit -- the SCS twin of
tests/test_gpu_parity.py::test_fused_search_equals_oracle_on_same_evaluations.
"""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def image_digest(img):
    return hashlib.blake2b(np.ascontiguousarray(img, dtype=np.float32).tobytes(), digest_size=8).digest()


def _mix64(x):
    """murmur3's finaliser on uint64 arrays (wrap-around arithmetic), the twin of mix64 in scs_search.hip."""
    x = x.astype(np.uint64).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xff51afd7ed558ccd)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xc4ceb9fe1a85ec53)
        x ^= x >> np.uint64(33)
    return x


def image_mix_digest(img):
    """The 128-bit digest the persistent kernel records of a leaf's float32 planes (image_hash_wave, scs_search.hip):
    order-free sums of per-element mixes of (flat NCHW index, float bits).  Returns uint64 [2] (hi, lo)."""
    bits = np.ascontiguousarray(img, dtype=np.float32).reshape(-1).view(np.uint32).astype(np.uint64)
    v = (np.arange(len(bits), dtype=np.uint64) << np.uint64(32)) | bits
    with np.errstate(over="ignore"):
        a = _mix64(v ^ np.uint64(0x9e3779b97f4a7c15)).sum(dtype=np.uint64)
        b = _mix64(v * np.uint64(0xd6e8feb86659fd93) + np.uint64(0x2545f4914f6cdd1d)).sum(dtype=np.uint64)
        hi = _mix64(np.array([a ^ (b >> np.uint64(7))], np.uint64))[0]
        lo = _mix64(np.array([b ^ (a << np.uint64(9))], np.uint64))[0]
    return np.array([hi, lo], np.uint64)


class LeafRecorder:
    def __init__(self, evaluator, games=None):
        """games: the game indices to record (None: all)."""
        self.evaluator, self.games = evaluator, (None if games is None else set(int(g) for g in games))
        self.records = {}                      # game -> list of (digest, probs float32 [A], value float32)

    def __call__(self, images, leaf_game):
        probs, values = self.evaluator(images)
        lg = leaf_game.cpu().numpy()
        keep = [i for i, g in enumerate(lg) if self.games is None or int(g) in self.games]
        if keep:
            idx = images.new_tensor(keep, dtype=int)
            im, pr, va = images[idx].cpu().numpy(), probs[idx].cpu().numpy(), values[idx].cpu().numpy()
            for j, i in enumerate(keep):
                self.records.setdefault(int(lg[i]), []).append((image_digest(im[j]), pr[j].copy(), np.float32(va[j])))
        return probs, values

    def arrays(self, g):
        rec = self.records[g]
        return (np.frombuffer(b"".join(r[0] for r in rec), np.uint8).reshape(-1, 8).copy(),
                np.stack([r[1] for r in rec]), np.array([r[2] for r in rec], np.float32))


def replay_game(args):
    """(config path, search config, seed, training, digests [n,8], probs [n,A], values [n], max_moves) -> trace dict.
    Top-level so it can run in a worker process (CPU only: nothing here touches the GPU)."""
    config_path, search, seed, training, digests, probs, values, max_moves = args
    from oracle import search as osearch
    from oracle.scs import ScsConfig, ScsGame
    cfg = ScsConfig(config_path)
    game = ScsGame(cfg)
    cursor = [0]

    def ev(g):
        i = cursor[0]
        if i >= len(values):
            raise AssertionError("the oracle asks for more evaluations than the device search used")
        if digests.dtype == np.uint64:           # recorded by the persistent kernel itself (nz_scs_search_record)
            same = np.array_equal(image_mix_digest(g.state_image()[0]), digests[i])
        else:
            same = image_digest(g.state_image()[0]) == digests[i].tobytes()
        if not same:
            raise AssertionError(f"leaf {i}: the oracle's leaf image is not the one the device evaluated")
        cursor[0] = i + 1
        return probs[i], values[i]

    explorer = osearch.Explorer(search, training, np.random.RandomState(int(seed)))
    root = osearch.Node(0)
    trace = []
    while not game.is_terminal() and (not max_moves or len(trace) < max_moves):
        action, chosen, bias = explorer.run_mcts(game, ev, root)
        trace.append({"action": int(action), "root_visits": int(root.visit_count), "root_value_sum": float(root.value_sum),
                      "bias": float(bias),
                      "child_actions": [c.action for c in root.children],
                      "child_visits": [c.visit_count for c in root.children],
                      "child_priors": [float(c.prior) for c in root.children],
                      "child_value_sums": [float(c.value_sum) for c in root.children]})
        game.step_index(action)
        root = chosen
    return {"trace": trace, "length": game.length, "terminal": bool(game.is_terminal()),
            "terminal_value": game.terminal_value if game.is_terminal() else None,
            "evaluations_used": cursor[0], "evaluations_recorded": len(values)}


def replay_games(jobs, workers=None):
    """Replay several games, in worker processes when there is more than one (spawned: the parent holds a GPU)."""
    if len(jobs) <= 1 or workers == 1:
        return [replay_game(j) for j in jobs]
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    workers = workers or min(len(jobs), max(1, (os.cpu_count() or 2) - 1), 12)
    with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(replay_game, jobs))


def assert_trace_equals_device(r, g, out, label=""):
    """Every root statistic of game g of a device export `r` equals the oracle replay `out`, bit for bit."""
    trace = out["trace"]
    for m, mv in enumerate(trace):
        k = len(mv["child_actions"])
        where = (label, g, m)
        assert r["actions"][g, m] == mv["action"], where
        assert r["tree_size"][g, m] == mv["root_visits"] and r["n_children"][g, m] == k, where
        assert r["bias"][g, m] == mv["bias"] and r["root_value_sum"][g, m] == mv["root_value_sum"], where
        assert r["child_action"][g, m, :k].tolist() == mv["child_actions"], where
        assert r["child_visit"][g, m, :k].tolist() == mv["child_visits"], where
        assert r["child_prior"][g, m, :k].tolist() == mv["child_priors"], where
        assert r["child_value_sum"][g, m, :k].tolist() == mv["child_value_sums"], where
    return len(trace)
