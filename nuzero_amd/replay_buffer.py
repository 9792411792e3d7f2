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

    def save_to_file(self, file_path, step=None):
        torch.save({"buffer": self.buffer, "n_games": self.n_games}, file_path)

    def load_from_file(self, file_path, step=None):
        d = torch.load(file_path, weights_only=False)   # a file this class wrote
        self.buffer, self.n_games = d["buffer"], d["n_games"]
