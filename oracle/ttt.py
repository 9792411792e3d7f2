"""Oracle: Tic_Tac_Toe rules (test infrastructure only).

Restates /root/reference/Games/Tic_Tac_Toe/tic_tac_toe.py.  The board is a flat
list of nine cells (0 empty, 1 player one, 2 player two); action index =
row*3+col, which is what ``Game.get_action_coords`` (Games/Game.py:96-98) yields
for ``action_space_shape == (1,3,3)`` (tic_tac_toe.py:30-31).
"""
import numpy as np

NUM_ACTIONS = 9
STATE_SHAPE = (2, 3, 3)

_LINES = ((0, 1, 2), (3, 4, 5), (6, 7, 8),
          (0, 3, 6), (1, 4, 7), (2, 5, 8),
          (0, 4, 8), (2, 4, 6))


class TicTacToe:
    """Same duck-typed surface the Explorer/Gamer loop uses."""

    num_actions = NUM_ACTIONS

    def __init__(self):
        self.board = [0] * 9
        self.player = 1            # tic_tac_toe.py:27 -- players are 1 and 2
        self.length = 0
        self.terminal = False
        self.terminal_value = 0
        self.state_history = []
        self.child_policy = []

    # ---- rules -----------------------------------------------------------
    def get_current_player(self):
        return self.player

    def get_length(self):
        return self.length

    def get_num_actions(self):
        return NUM_ACTIONS

    def is_terminal(self):
        return self.terminal

    def get_terminal_value(self):
        return self.terminal_value

    def possible_actions(self):
        """tic_tac_toe.py:121-129: float64 ones where the cell is empty."""
        return np.array([1.0 if c == 0 else 0.0 for c in self.board],
                        dtype=np.float64).reshape(3, 3)

    def step_index(self, action_i):
        """tic_tac_toe.py:131-133,161-167 with coords = unravel(action_i,(1,3,3))."""
        self.board[action_i] = self.player
        self.length += 1
        self._check_terminal()
        self.player = (self.length % 2) + 1
        return self.terminal

    def _check_terminal(self):
        """tic_tac_toe.py:198-262.  A line of three wins (+1 for player one, -1
        for player two, player one checked first); otherwise a full board is a
        draw.  A winning ninth move is a win, not a draw (SURVEY.md rule 21)."""
        b = self.board
        value = 0
        done = False
        if any(b[i] == 1 and b[j] == 1 and b[k] == 1 for i, j, k in _LINES):
            value, done = 1, True
        elif any(b[i] == 2 and b[j] == 2 and b[k] == 2 for i, j, k in _LINES):
            value, done = -1, True
        if self.length == 9:
            done = True
        if done:
            self.terminal_value = value
            self.terminal = True

    def state_image(self):
        """tic_tac_toe.py:135-159: [1,2,3,3] float32 = (P1 stones, P2 stones);
        no side-to-move plane."""
        img = np.zeros((1, 2, 3, 3), dtype=np.float32)
        for a, c in enumerate(self.board):
            if c:
                img[0, c - 1, a // 3, a % 3] = 1.0
        return img

    generate_network_input = state_image

    def shallow_clone(self):
        """tic_tac_toe.py:267-273: board, player and length only."""
        g = TicTacToe()
        g.board = list(self.board)
        g.player = self.player
        g.length = self.length
        return g

    # ---- bookkeeping used by the Gamer loop --------------------------------
    def store_state(self, state):
        self.state_history.append(state)

    def store_search_statistics(self, root):
        """tic_tac_toe.py:177-182: visit fraction per action, 0 for non-children."""
        total = sum(c.visit_count for c in root.children)
        by_action = {c.action: c.visit_count for c in root.children}
        self.child_policy.append(
            [by_action[a] / total if a in by_action else 0 for a in range(NUM_ACTIONS)])

    def get_state_from_history(self, i):
        return self.state_history[i]

    def make_target(self, i):
        """tic_tac_toe.py:184-190."""
        return (self.terminal_value, self.child_policy[i])

    def code(self):
        """Base-3 position code (cell a has weight 3**a); not in the reference --
        used as the key of the table evaluator."""
        k = 0
        for a in range(8, -1, -1):
            k = k * 3 + self.board[a]
        return k


def board_from_code(code):
    b = []
    for _ in range(9):
        b.append(code % 3)
        code //= 3
    return b


def reachable_positions():
    """All positions reachable from the empty board by legal play (terminal
    ones included), as a sorted list of base-3 codes."""
    seen = set()
    stack = [TicTacToe()]
    while stack:
        g = stack.pop()
        k = g.code()
        if k in seen:
            continue
        seen.add(k)
        if g.terminal:
            continue
        for a in range(9):
            if g.board[a] == 0:
                h = g.shallow_clone()
                h.step_index(a)
                stack.append(h)
    return sorted(seen)
