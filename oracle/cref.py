"""Python driver of the C oracle (oracle/c/mcts_ref.c) -- test infrastructure only.

Plays a batch of games with a table evaluator; randomness comes from numpy
RandomState objects, one per game, in the reference's call order."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libmcts_ref.so")


def _lib():
    if not os.path.exists(_LIB):
        import runpy
        runpy.run_path(os.path.join(_HERE, "c", "build.py"), run_name="__main__")
    lib = ctypes.CDLL(_LIB)
    lib.oc_create.restype = ctypes.c_void_p
    lib.oc_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                              ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                              ctypes.c_int, ctypes.c_void_p]
    lib.oc_destroy.argtypes = [ctypes.c_void_p]
    lib.oc_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.oc_move.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.oc_export.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 11
    return lib


def play_games(table, search_config, seeds, training=True, negate_player=2):
    lib = _lib()
    table = np.ascontiguousarray(table, np.float32)
    sim, uct, ex = search_config["Simulation"], search_config["UCT"], search_config["Exploration"]
    G = len(seeds)
    h = lib.oc_create(G, int(sim["mcts_simulations"]), float(uct["pb_c_base"]), float(uct["pb_c_init"]),
                      float(ex["value_factor"]), float(ex["root_exploration_fraction"]),
                      int(ex["number_of_softmax_moves"]), float(ex["epsilon_softmax_exploration"]),
                      float(ex["epsilon_random_exploration"]), int(training), negate_player,
                      table.ctypes.data_as(ctypes.c_void_p))
    rngs = [np.random.RandomState(int(s)) for s in seeds]
    alive = np.zeros(G, np.int32)
    nchild = np.zeros(G, np.int32)
    for move in range(9):
        lib.oc_state(h, alive.ctypes.data, nchild.ctypes.data)
        if not alive.any():
            break
        noise = np.zeros((G, 9), np.float64)
        uni = np.zeros((G, 3), np.float64)
        if training:
            for g in np.nonzero(alive)[0]:
                rs = rngs[g]
                n = int(nchild[g])
                noise[g, :n] = rs.gamma(ex["root_dist_alpha"], ex["root_dist_beta"], n)
                if move < ex["number_of_softmax_moves"]:
                    uni[g, 2] = rs.random_sample()
                else:
                    u1, u2 = rs.random_sample(), rs.random_sample()
                    uni[g, 0], uni[g, 1] = u1, u2
                    if u1 < ex["epsilon_softmax_exploration"] or u2 < ex["epsilon_random_exploration"]:
                        uni[g, 2] = rs.random_sample()
        lib.oc_move(h, noise.ctypes.data, uni.ctypes.data)
    out = {"visits": np.zeros((G, 9, 9), np.int32), "actions": np.zeros((G, 9), np.int32),
           "lengths": np.zeros(G, np.int32), "outcomes": np.zeros(G, np.int32),
           "tree_size": np.zeros((G, 9), np.int32), "n_children": np.zeros((G, 9), np.int32),
           "bias": np.zeros((G, 9)), "child_prior": np.zeros((G, 9, 9)), "child_value_sum": np.zeros((G, 9, 9)),
           "root_value_sum": np.zeros((G, 9))}
    counters = np.zeros(2, np.int64)
    lib.oc_export(h, *[out[k].ctypes.data for k in ("visits", "actions", "lengths", "outcomes", "tree_size",
                                                    "n_children", "bias", "child_prior", "child_value_sum",
                                                    "root_value_sum")], counters.ctypes.data)
    lib.oc_destroy(h)
    out["simulations"], out["expansions"] = int(counters[0]), int(counters[1])
    return out
