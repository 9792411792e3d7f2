"""CPU oracle for the self-play hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference's Explorer/Gamer MCTS loop,
the Tic_Tac_Toe rules and the square-conv RecurrentNet forward.  It exists to
check the HIP path; nothing in ``nuzero_amd/`` may import it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imported the genuine
reference (``/root/reference``: Search/Explorer.py, Search/Node.py,
Games/Tic_Tac_Toe/tic_tac_toe.py, Neural_Networks/*) in the build container and
wrote the vectors under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function here against them.

Modules
  ttt.py      Tic_Tac_Toe rules           (Games/Tic_Tac_Toe/tic_tac_toe.py)
  search.py   Node / Explorer / Gamer loop (Search/Node.py, Search/Explorer.py,
                                            Training/Gamer.py:52-92)
  net.py      RecurrentNet forward, torch fp32 (Neural_Networks/Architectures/
                                            RecurrentNet.py, blocks.py)
  c/          plain-C restatement of search.py + ttt.py for full-size checks
"""
