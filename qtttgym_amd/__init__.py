"""qtttgym_amd — MI355X-native vectorised Quantum Tic-Tac-Toe environment.

Exports the reference package's four names (qtttgym/__init__.py:1-4) plus `VecEnv` and the 36-action
indexing L3 callers share (mcts.py:339-350)."""
from .vec_env import VecEnv
from .board import Board, QEvalClassic, displayBoard
from .env import Env
from .actions import ind2move, move2ind
from ._native import recommended_env, retire_mailbox

__all__ = ["Board", "QEvalClassic", "displayBoard", "Env", "VecEnv", "ind2move", "move2ind", "recommended_env",
           "retire_mailbox"]
