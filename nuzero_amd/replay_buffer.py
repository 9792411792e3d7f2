"""ReplayBuffer with the reference's surface (Training/ReplayBuffer.py:10-107),
without Ray: a plain object owned by the process that trains.

Entries are the reference's tuples ``(state [1,C,H,W] float32 tensor,
(value, policy list), game_index)``, so Training/AlphaZero.py's
batch_update_weights (AlphaZero.py:836-889) consumes them unchanged.
"""
import random

import numpy as np
import torch


class ReplayBuffer:
    def __init__(self, window_size, batch_size):
        self.window_size = window_size
        self.batch_size = batch_size
        self.buffer = []
        self.n_games = 0
        self.full = False
        # checkpoints by training step (ReplayBuffer.py:20-22)
        self.step_to_size_map = {}
        self.allow_partial_loading = True

    def save_game(self, game, game_index):
        """ReplayBuffer.py:24-36: once the window holds window_size games, one
        oldest position is dropped per position added."""
        if self.n_games >= self.window_size:
            self.full = True
        else:
            self.full = False
            self.n_games += 1
        for i in range(len(game.state_history)):
            entry = (game.get_state_from_history(i), game.make_target(i), game_index)
            if self.full:
                self.buffer.pop(0)
            self.buffer.append(entry)

    def save_games(self, games, game_index):
        for g in games:
            self.save_game(g, game_index)

    def shuffle(self):
        random.shuffle(self.buffer)

    def get_slice(self, start_index, last_index):
        return self.buffer[start_index:last_index]

    def get_sample(self, batch_size, replace, probs):
        args = [len(self.buffer), batch_size, replace] if probs == [] else [len(self.buffer), batch_size, replace, probs]
        return [self.buffer[i] for i in np.random.choice(*args)]

    def get_buffer(self):
        return self.buffer

    def len(self):
        return len(self.buffer)

    def played_games(self):
        return self.n_games

    def save_to_file(self, file_path, step):
        """ReplayBuffer.py:64-79: the reference's checkpoint layout {'buffer', 'map', 'partial_loading'}."""
        self.step_to_size_map[step] = (self.len(), self.played_games())
        if self.full:
            # once old entries are being dropped an older, shorter prefix of the buffer no longer exists
            self.allow_partial_loading = False
        torch.save({"buffer": self.buffer, "map": self.step_to_size_map, "partial_loading": self.allow_partial_loading},
                   file_path)

    def load_from_file(self, file_path, step):
        """ReplayBuffer.py:81-107 (files written by the reference load as well)."""
        self.buffer, self.n_games, self.step_to_size_map, self.allow_partial_loading = \
            load_reference_checkpoint(file_path, step)
        self.full = self.n_games >= self.window_size


def load_reference_checkpoint(file_path, step):
    """Read a replay-buffer checkpoint in the reference's layout and apply its loading rule (ReplayBuffer.py:81-107):
    with partial loading allowed, the buffer as it was at `step` (`buffer[:buffer_len + 1]`, the reference's slice);
    otherwise the whole latest buffer.  Returns (buffer, n_games, map, partial_loading).  The file is read with
    torch.load(weights_only=True): its entries are tensors, tuples, lists and numbers, nothing is unpickled freely."""
    checkpoint = torch.load(file_path, weights_only=True)
    buffer, size_map, partial = checkpoint["buffer"], checkpoint["map"], checkpoint["partial_loading"]
    if partial:
        if step not in size_map:
            raise Exception("Could not load the replay buffer checkpoint for that iteration number.")
        buffer_len, num_games = size_map[step]
        return buffer[:buffer_len + 1], num_games, size_map, partial
    latest_step, (buffer_len, num_games) = list(size_map.items())[-1]
    if step != latest_step:
        print("Partial loading is no longer possible.")
        print("Loading the latest buffer instead.")
    return buffer, num_games, size_map, partial
