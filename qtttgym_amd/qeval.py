"""`qtttgym.qeval` by module path (qtttgym/qeval.py:4): the class lives in board.py beside the façade that decides whether
its `eval` runs on the device."""
from .board import QEvalClassic

__all__ = ["QEvalClassic"]
