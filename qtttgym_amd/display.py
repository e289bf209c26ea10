"""`qtttgym.display` by module path (qtttgym/display.py:4; strat_eval.py:44,58 name it as `qtttgym.display.displayBoard`)."""
from .board import displayBoard

__all__ = ["displayBoard"]
