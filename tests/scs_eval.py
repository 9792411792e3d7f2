"""A deterministic stand-in evaluator for SCS search tests: (post-softmax probabilities, value)
as a pure function of the state image.  Used on both sides of every comparison (reference,
oracle, device), so no network arithmetic is involved."""
import math

import numpy as np


def checksum_weights(n):
    i = np.arange(n, dtype=np.int64)
    return ((i * 2654435761) % 1000003).astype(np.float64) / 1000003.0


_W = {}


def evaluate_image(img, num_actions):
    """img: float32 [C, R, Cc] -> (probs float32 [num_actions], value np.float32)."""
    flat = np.asarray(img, np.float32).reshape(-1)
    w = _W.get(flat.size)
    if w is None:
        w = _W[flat.size] = checksum_weights(flat.size)
    s = float(np.sum(flat.astype(np.float64) * w))
    k = int(abs(s) * 977.0) % 17
    p = (1.0 + ((np.arange(num_actions, dtype=np.int64) * 31 + k) % 17)).astype(np.float32)
    p = p / np.sum(p)
    return p, np.float32(math.sin(s))
