"""qtttgym_amd — MI355X-native vectorised Quantum Tic-Tac-Toe environment.

Exports the reference package's four names (qtttgym/__init__.py:1-4) plus `VecEnv`."""
from .vec_env import VecEnv
from .board import Board, QEvalClassic, displayBoard
from .env import Env

__all__ = ["Board", "QEvalClassic", "displayBoard", "Env", "VecEnv"]
