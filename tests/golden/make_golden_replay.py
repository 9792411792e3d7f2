#!/usr/bin/env python3
"""Golden vectors for the replay buffer and the loss functions, from the GENUINE reference classes.

Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_replay.py

Imported from /root/reference (read-only, nothing copied):
  Training.ReplayBuffer.ReplayBuffer   with an inert `ray` stand-in whose remote(**kw) returns the class unchanged
                                       (the decorator only registers the class with Ray, ReplayBuffer.py:11)
  Utils.Functions.loss_functions       KLDivergence, MSError, SquaredError, AbsoluteError (imports cleanly)
`AlphaZero.calculate_loss` (Training/AlphaZero.py:891-955) cannot be imported (Ray, ruamel, more_itertools, progress are
absent): its 20-line per-sample loop is driven here around the imported loss functions and torch's own
nn.CrossEntropyLoss(label_smoothing=0.02) (AlphaZero.py:327), as make_golden.py does for Gamer's move loop; the late_heavy
probabilities (AlphaZero.py:777-795) and the bucketing by game index (AlphaZero.py:846-852, more_itertools.bucket =
stable grouping, keys ascending) are restated the same way.

Outputs: replay_kat.json (window / eviction / shuffle / slice / sample traces as (game, move) identities),
         loss_kat.npz (inputs, per-configuration losses and autograd gradients).
"""
import json
import math
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

_ray = types.ModuleType("ray")
_ray.remote = lambda *a, **kw: (lambda cls: cls)
sys.modules["ray"] = _ray

import torch  # noqa: E402
from torch import nn  # noqa: E402

from Training.ReplayBuffer import ReplayBuffer  # noqa: E402
from Utils.Functions.loss_functions import KLDivergence, MSError, SquaredError, AbsoluteError  # noqa: E402

torch.set_num_threads(1)


class FakeGame:
    """What ReplayBuffer.save_game reads (ReplayBuffer.py:31-33): state_history, get_state_from_history, make_target."""

    def __init__(self, gid, length, num_actions, rs):
        self.gid = gid
        self.state_history = [torch.tensor([[float(gid), float(m)]]) for m in range(length)]
        self.value = int(rs.randint(-1, 2))
        self.policies = []
        for m in range(length):
            v = rs.randint(0, 50, size=num_actions) * (rs.random_sample(num_actions) < 0.6)
            if v.sum() == 0:
                v[rs.randint(num_actions)] = 7
            total = int(v.sum())
            self.policies.append([int(x) / total for x in v])
        self.visits = None

    def get_state_from_history(self, i):
        return self.state_history[i]

    def make_target(self, i):
        return (self.value, self.policies[i])


def ident(entry):
    state, (value, policy), game_index = entry
    return [int(state[0, 0]), int(state[0, 1]), int(game_index)]


def late_heavy_probs(num_positions):                 # AlphaZero.py:780-795
    probs = []
    variation = 0.5
    offset = (1 - variation) / 2
    fraction = variation / num_positions
    total = offset
    for _ in range(num_positions):
        total += fraction
        probs.append(total)
    total_sum = sum(probs)
    return [p / total_sum for p in probs]


def gen_replay():
    cases = {}
    for name, window, n_games, max_len, seed in (("small_window", 5, 14, 6, 1), ("never_full", 50, 9, 9, 2),
                                                 ("window_1", 1, 4, 5, 3), ("long", 20, 60, 9, 4)):
        rs = np.random.RandomState(seed)
        rb = ReplayBuffer(window, 4)
        ops = []
        lengths = [int(rs.randint(1, max_len + 1)) for _ in range(n_games)]
        game_types = [int(rs.randint(0, 3)) for _ in range(n_games)]
        values = []
        for g in range(n_games):
            game = FakeGame(g, lengths[g], 9, rs)
            values.append(game.value)
            rb.save_game(game, game_types[g])
            ops.append({"op": "save", "game": g, "len": rb.len(), "played": rb.played_games(), "full": bool(rb.full),
                        "order": [ident(e)[:2] for e in rb.get_buffer()]})
            if g % 5 == 4:
                random.seed(1000 + g)
                rb.shuffle()
                ops.append({"op": "shuffle", "seed": 1000 + g, "order": [ident(e)[:2] for e in rb.get_buffer()]})
                a = min(2, rb.len())
                b = min(a + 4, rb.len())
                ops.append({"op": "slice", "start": a, "stop": b, "got": [ident(e) for e in rb.get_slice(a, b)]})
            if g % 7 == 6:
                for replace, late in ((True, False), (False, False), (True, True)):
                    bs = min(4, rb.len())
                    probs = late_heavy_probs(rb.len()) if late else []
                    np.random.seed(2000 + g)
                    batch = rb.get_sample(bs, replace, probs)
                    ops.append({"op": "sample", "seed": 2000 + g, "batch_size": bs, "replace": replace, "late_heavy": late,
                                "got": [ident(e) for e in batch]})
        # batch assembly of the last sample, as batch_update_weights groups it (AlphaZero.py:846-852)
        np.random.seed(77)
        batch = rb.get_sample(min(8, rb.len()), True, [])
        keys = sorted(set(e[2] for e in batch))
        grouped = [[ident(e) for e in batch if e[2] == k] for k in keys]
        ops.append({"op": "bucket", "seed": 77, "batch_size": min(8, rb.len()), "keys": keys, "groups": grouped})
        cases[name] = {"window": window, "lengths": lengths, "game_types": game_types, "values": values, "seed": seed,
                       "ops": ops}
    # the contents of one buffer, position by position (policy as the float32 tensor the trainer makes of it)
    rs = np.random.RandomState(9)
    rb = ReplayBuffer(3, 4)
    games = [FakeGame(g, int(rs.randint(2, 6)), 9, rs) for g in range(5)]
    for g in games:
        rb.save_game(g, g.gid % 2)
    content = [{"id": ident(e), "value": int(e[1][0]),
                "policy_f32": torch.tensor(e[1][1]).numpy().astype(np.float32).tolist(),
                "policy_f64": [float(x) for x in e[1][1]]} for e in rb.get_buffer()]
    cases["content"] = {"window": 3, "lengths": [len(g.state_history) for g in games], "seed": 9, "content": content,
                        "visit_like": [[[round(p * 1e6) for p in pol] for pol in g.policies] for g in games]}
    with open(os.path.join(HERE, "replay_kat.json"), "w") as f:
        json.dump(cases, f)
    return {k: len(v.get("ops", [])) for k, v in cases.items()}


def calculate_loss(outputs, targets, batch_size, policy_loss_function, value_loss_function, normalize_policy):
    """AlphaZero.calculate_loss (Training/AlphaZero.py:891-921), the loop as it stands there."""
    target_values, target_policies = list(zip(*targets))
    predicted_policies, predicted_values = outputs
    policy_loss = 0.0
    value_loss = 0.0
    for i in range(batch_size):
        target_policy = torch.tensor(target_policies[i])
        target_value = torch.tensor(target_values[i])
        predicted_value = predicted_values[i]
        predicted_policy = torch.flatten(predicted_policies[i])
        policy_loss += policy_loss_function(predicted_policy, target_policy)
        value_loss += value_loss_function(predicted_value, target_value)
    target_size = len(targets)
    if normalize_policy:
        policy_loss /= math.log(target_size)
    value_loss /= batch_size
    policy_loss /= batch_size
    return value_loss, policy_loss, policy_loss + value_loss


def gen_loss():
    out = {}
    meta = {}
    for name, B, planes, rows, cols, seed in (("ttt", 37, 1, 3, 3, 5), ("scs", 12, 21, 5, 5, 6), ("one", 1, 1, 3, 3, 7)):
        rs = np.random.RandomState(seed)
        A = planes * rows * cols
        logits = (rs.standard_normal((B, planes, rows, cols)) * 2.0).astype(np.float32)
        values = np.tanh(rs.standard_normal((B, 1))).astype(np.float32)
        targets = []
        for i in range(B):
            k = int(rs.randint(1, min(A, 12) + 1))
            idx = rs.choice(A, size=k, replace=False)
            visits = rs.randint(1, 60, size=k)
            pol = [0] * A
            for a, v in zip(idx, visits):
                pol[int(a)] = int(v) / int(visits.sum())
            targets.append((int(rs.randint(-1, 2)), pol))
        out[f"{name}_logits"], out[f"{name}_values"] = logits, values
        out[f"{name}_target_values"] = np.array([t[0] for t in targets], np.int32)
        out[f"{name}_target_policies"] = np.array([t[1] for t in targets], np.float64)
        for pname, pfun, norm in (("ce", nn.CrossEntropyLoss(label_smoothing=0.02), False),
                                  ("ce_norm", nn.CrossEntropyLoss(label_smoothing=0.02), True),
                                  ("kld", KLDivergence, False), ("mse", MSError, False)):
            for vname, vfun in (("se", SquaredError), ("ae", AbsoluteError)):
                if pname == "ce_norm" and B == 1:
                    continue                 # log(1) = 0: the reference divides by zero there
                lg = torch.tensor(logits, requires_grad=True)
                vl = torch.tensor(values, requires_grad=True)
                v_loss, p_loss, c_loss = calculate_loss((lg, vl), targets, B, pfun, vfun, norm)
                c_loss.backward()
                key = f"{name}_{pname}_{vname}"
                out[key + "_losses"] = np.array([float(v_loss), float(p_loss), float(c_loss)], np.float64)
                out[key + "_dlogits"] = lg.grad.numpy().copy()
                out[key + "_dvalues"] = vl.grad.numpy().copy()
        meta[name] = {"batch": B, "actions": A}
    np.savez_compressed(os.path.join(HERE, "loss_kat.npz"), **out)
    return meta


if __name__ == "__main__":
    print(json.dumps({"replay": gen_replay(), "loss": gen_loss(), "torch": torch.__version__, "numpy": np.__version__}))
