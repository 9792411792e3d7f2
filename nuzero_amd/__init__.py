"""MI355X-native self-play engine behind NuZero's Gamer / Game / Network_Manager /
ReplayBuffer surface.

Only the self-play hot path lives here: batched MCTS over
thousands of game trees with policy/value inference fused in, as hand-written
HIP kernels for gfx950 reached through the C ABI of include/nuzero_amd.h.
Importing a submodule that needs the shared library raises if it has not been
built (``python -m nuzero_amd.build``); there is no CPU fallback.
"""
__version__ = "0.1"
