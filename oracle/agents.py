"""Oracle: evaluation-time MCTS agents and the match loop (test infrastructure only).

Restates Testing/Agents/Generic/MctsAgent.py:14-45 and the core of
Testing/Tester.py:46-121 (Test_using_agents) on top of oracle/search.py."""
from .search import Explorer, Node


class MctsAgentRef:
    def __init__(self, search_config, evaluator):
        self.explorer = Explorer(search_config, False)          # training=False: no noise, max action
        self.keep_subtree = search_config["Simulation"]["keep_subtree"]
        self.evaluator = evaluator
        self.root = Node(0)

    def new_game(self):
        self.root = Node(0)

    def choose_action(self, game):                              # MctsAgent.py:28-33
        action, chosen, _ = self.explorer.run_mcts(game, self.evaluator, self.root)
        if self.keep_subtree:
            self.root = chosen
        return action

    def update_subtree(self, game, action):                     # MctsAgent.py:35-39
        self.explorer.run_mcts(game, self.evaluator, self.root)
        self.root = self.root.child(action)


def play_match(game, p1, p2):
    """Tester.py:62-118: the mover chooses; an MCTS opponent that keeps its subtree searches the
    same position and follows the move; then the game steps.  Returns the action list."""
    p1.new_game()
    p2.new_game()
    actions = []
    while True:
        cur, opp = (p1, p2) if game.get_current_player() == 1 else (p2, p1)
        a = cur.choose_action(game)
        if isinstance(opp, MctsAgentRef) and opp.keep_subtree:
            opp.update_subtree(game, a)
        actions.append(a)
        game.step_index(a)
        if game.is_terminal():
            return actions
