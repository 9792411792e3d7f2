"""The reference's search-config schema (Configs/Search/*.yaml) -> nz_search_cfg.

Explorer reads nested dict keys (Search/Explorer.py:48,74,79-80,105-106,122,
202-205); the same nested dict is accepted here unchanged.
"""
import copy

LEGACY_TTT_SEARCH_CONFIG = {
    # Games/Tic_Tac_Toe/models/best_ttt_config/search_config_copy.ini -- the
    # configuration BASELINE.json's Tic-Tac-Toe workloads are quoted on
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


def legacy_ttt_search_config(mcts_simulations=100):
    cfg = copy.deepcopy(LEGACY_TTT_SEARCH_CONFIG)
    cfg["Simulation"]["mcts_simulations"] = mcts_simulations
    return cfg


def to_struct(search_config, training):
    from ._lib import SearchCfg
    sim, uct, ex = search_config["Simulation"], search_config["UCT"], search_config["Exploration"]
    dist = ex.get("root_exploration_distribution", "gamma")
    if dist != "gamma":
        raise ValueError(f"root_exploration_distribution {dist!r}: only 'gamma' exists (Explorer.py:208)")
    return SearchCfg(
        mcts_simulations=int(sim["mcts_simulations"]),
        keep_subtree=int(bool(sim["keep_subtree"])),
        pb_c_base=float(uct["pb_c_base"]),
        pb_c_init=float(uct["pb_c_init"]),
        number_of_softmax_moves=int(ex["number_of_softmax_moves"]),
        training=int(bool(training)),
        epsilon_softmax_exploration=float(ex["epsilon_softmax_exploration"]),
        epsilon_random_exploration=float(ex["epsilon_random_exploration"]),
        value_factor=float(ex["value_factor"]),
        root_exploration_fraction=float(ex["root_exploration_fraction"]),
        root_dist_alpha=float(ex["root_dist_alpha"]),
        root_dist_beta=float(ex["root_dist_beta"]),
    )
