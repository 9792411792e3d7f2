"""Oracle: AlphaZero MCTS + self-play move loop (test infrastructure only).

Restates /root/reference/Search/Node.py, Search/Explorer.py and the loop of
Training/Gamer.py:52-92.  Numeric types follow SURVEY.md appendix A: tree
arithmetic is Python float (IEEE double), softmax output is float32, the
masked/renormalised priors are numpy float64 (float32 probs x float64 mask).

Differences from the reference that do not change results:
  * children are a list in ascending action order (the reference's dict is
    filled in ascending action order, Explorer.py:177-179, so iteration order is
    the same);
  * the random stream is an explicit ``np.random.RandomState`` per game instead
    of the legacy global one (same MT19937 stream, SURVEY.md rule 14/17);
  * the network is reached through an ``evaluator(game) -> (probs_f32, value)``
    callable.  ``net_evaluator`` is the reference's path (inference + scipy
    softmax, Explorer.py:158-162); ``table_evaluator`` looks post-softmax
    probabilities up by position, which is what the reference does on a cache
    hit (Explorer.py:147-149).
"""
import math

import numpy as np
from scipy.special import softmax

DEFAULT_SEARCH_CONFIG = {
    # Games/Tic_Tac_Toe/models/best_ttt_config/search_config_copy.ini
    "Simulation": {"mcts_simulations": 100, "keep_subtree": True},
    "UCT": {"pb_c_base": 5000, "pb_c_init": 1.15},
    "Exploration": {
        "number_of_softmax_moves": 0,
        "epsilon_softmax_exploration": 0,
        "epsilon_random_exploration": 0,
        "value_factor": 1,
        "root_exploration_distribution": "gamma",
        "root_exploration_fraction": 0.2,
        "root_dist_alpha": 0.15,
        "root_dist_beta": 1,
    },
}


class Node:
    """Search/Node.py:3-32."""
    __slots__ = ("visit_count", "prior", "value_sum", "terminal_value",
                 "children", "to_play", "action")

    def __init__(self, prior, action=-1):
        self.visit_count = 0
        self.prior = prior
        self.value_sum = 0
        self.terminal_value = None
        self.children = []      # ascending action order
        self.to_play = -1
        self.action = action

    def expanded(self):
        return len(self.children) > 0

    def value(self):
        if self.visit_count == 0:
            return 0.0
        return self.value_sum / self.visit_count

    def child(self, action):
        for c in self.children:
            if c.action == action:
                return c
        raise KeyError(action)


class Counters:
    """Work counters for the throughput metrics of SURVEY.md section 8(d)."""

    def __init__(self):
        self.simulations = 0
        self.expansions = 0     # evaluate() calls on non-terminal leaves


class Explorer:
    """Search/Explorer.py:35-210."""

    def __init__(self, search_config, training, rng=None, negate_player=2):
        self.config = search_config
        self.training = training
        self.rng = rng if rng is not None else np.random.RandomState()
        # Explorer.py:124 negates Q iff parent.to_play == 2 (SURVEY.md rule 6)
        self.negate_player = negate_player
        self.counters = Counters()

    # -- Explorer.py:40-67 ---------------------------------------------------
    def run_mcts(self, game, evaluator, root):
        if self.training:
            self.add_exploration_noise(root)
        for _ in range(self.config["Simulation"]["mcts_simulations"]):
            node = root
            scratch = game.shallow_clone()
            path = [node]
            while node.expanded():
                node = self.select_child(node)
                scratch.step_index(node.action)
                path.append(node)
            value = self.evaluate(node, scratch, evaluator)
            for n in path:                      # Explorer.py:132-135
                n.visit_count += 1
                n.value_sum += value
            self.counters.simulations += 1
        bias = self.exploration_bias(root)
        action = self.select_action(game, root)
        return action, root.child(action), bias

    # -- Explorer.py:99-130 --------------------------------------------------
    def exploration_bias(self, node):
        base = self.config["UCT"]["pb_c_base"]
        init = self.config["UCT"]["pb_c_init"]
        return math.log((node.visit_count + base + 1) / base) + init

    def score(self, parent, child):
        c = self.exploration_bias(parent)
        u = math.sqrt(parent.visit_count) / (child.visit_count + 1)
        conf = child.prior * u
        conf = conf * c
        q = child.value()
        if parent.to_play == self.negate_player:
            q = -q
        q = q * self.config["Exploration"]["value_factor"]
        return conf + q

    def select_child(self, parent):
        # max over (score, action) tuples: score ties go to the larger action
        best = None
        best_key = None
        for ch in parent.children:
            key = (self.score(parent, ch), ch.action)
            if best is None or key > best_key:
                best, best_key = ch, key
        return best

    # -- Explorer.py:137-181 -------------------------------------------------
    def evaluate(self, node, game, evaluator):
        node.to_play = game.get_current_player()
        if game.is_terminal():
            node.terminal_value = game.get_terminal_value()
            return node.terminal_value
        probs_f32, value = evaluator(game)
        value = float(value)                     # Explorer.py:162: predicted_value.item()
        self.counters.expansions += 1
        mask = game.possible_actions().flatten()
        probs = probs_f32.flatten() * mask
        total = np.sum(probs)
        if total == 0:
            probs += mask
            total = np.sum(probs)
        for a in range(game.get_num_actions()):
            if mask[a]:
                node.children.append(Node(probs[a] / total, a))
        return value

    # -- Explorer.py:201-210 -------------------------------------------------
    def add_exploration_noise(self, node):
        ex = self.config["Exploration"]
        frac = ex["root_exploration_fraction"]
        noise = self.rng.gamma(ex["root_dist_alpha"], ex["root_dist_beta"],
                               len(node.children))
        for ch, n in zip(node.children, noise):
            ch.prior = ch.prior * (1 - frac) + n * frac

    # -- Explorer.py:70-97,183-199 -------------------------------------------
    def select_action(self, game, root):
        ex = self.config["Exploration"]
        if not self.training:
            return self.max_action(root)
        if game.get_length() < ex["number_of_softmax_moves"]:
            return self.softmax_action(root)
        eps_softmax = self.rng.random_sample()
        eps_random = self.rng.random_sample()
        if eps_softmax < ex["epsilon_softmax_exploration"]:
            return self.softmax_action(root)
        if eps_random < ex["epsilon_random_exploration"]:
            mask = game.possible_actions().flatten()
            probs = mask / np.sum(mask)
            return int(self.rng.choice(game.get_num_actions(), p=probs))
        return self.max_action(root)

    @staticmethod
    def max_action(root):
        best = root.children[0]
        for ch in root.children[1:]:
            if ch.visit_count > best.visit_count:   # first maximum wins
                best = ch
        return best.action

    def softmax_action(self, root):
        counts = [ch.visit_count for ch in root.children]
        actions = [ch.action for ch in root.children]
        probs = np.asarray(softmax(counts), dtype=np.float64)
        probs /= np.sum(probs)
        return int(self.rng.choice(actions, p=probs))


# ---- evaluators ------------------------------------------------------------
def net_evaluator(net, iters):
    """Explorer.py:158-162: inference, scipy softmax over all logits, .item()."""
    def ev(game):
        logits, value = net.inference(game.generate_network_input(), iters)
        return softmax(logits), float(value.reshape(-1)[0])
    return ev


def table_evaluator(table):
    """``table[code] = 10 float32`` (9 post-softmax probabilities + value)."""
    def ev(game):
        row = table[game.code()]
        return row[:9].astype(np.float32), float(row[9])
    return ev


# ---- Gamer.play_game (Training/Gamer.py:52-92) -----------------------------
def play_game(game, evaluator, search_config, rng, training=True, trace=None,
              negate_player=2):
    explorer = Explorer(search_config, training, rng, negate_player)
    keep_subtree = search_config["Simulation"]["keep_subtree"]
    stats = {"number_of_moves": 0, "average_children": 0, "average_tree_size": 0,
             "final_tree_size": 0, "average_bias_value": 0, "final_bias_value": 0}
    root = Node(0)
    while not game.is_terminal():
        game.store_state(game.generate_network_input())
        action, chosen, bias = explorer.run_mcts(game, evaluator, root)
        tree_size = root.visit_count
        n_children = len(root.children)
        if trace is not None:
            trace.append({
                "action": action,
                "root_visits": root.visit_count,
                "root_value_sum": float(root.value_sum),
                "child_actions": [c.action for c in root.children],
                "child_visits": [c.visit_count for c in root.children],
                "child_priors": [float(c.prior) for c in root.children],
                "child_value_sums": [float(c.value_sum) for c in root.children],
            })
        game.step_index(action)
        game.store_search_statistics(root)
        if keep_subtree:
            root = chosen
        stats["average_children"] += n_children
        stats["average_tree_size"] += tree_size
        stats["final_tree_size"] = tree_size
        stats["average_bias_value"] += bias
        stats["final_bias_value"] = bias
    stats["number_of_moves"] = game.length
    stats["average_children"] /= game.length
    stats["average_tree_size"] /= game.length
    stats["average_bias_value"] /= game.length
    return stats, explorer.counters
