"""Compile the C oracle (test infrastructure) into oracle/_build/libmcts_ref.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_build")
LIB = os.path.join(OUT_DIR, "libmcts_ref.so")
SRC = os.path.join(HERE, "mcts_ref.c")


def build():
    os.makedirs(OUT_DIR, exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build())
