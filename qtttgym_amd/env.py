"""`Env` — the reference's single-board gymnasium-style surface (qtttgym/env.py:15-112),
returning the same Python types (dict observation with an aliasing `classical` list, float
reward incl. -0.0, bool terminated, False, {}), with the rules running in the HIP kernel via
the `Board` façade.  For N boards per call use `VecEnv`.

As in the reference (env.py:5-8,15) `Env` IS a `gymnasium.Env` with real gymnasium spaces whenever gymnasium is
importable (spaces.py); without it — the target image has none — it is a plain class with the declarative stand-ins."""
from .board import Board, QEvalClassic, displayBoard
from .spaces import GYM_ENV_BASE, reference_action_space, reference_observation_space


class Env(GYM_ENV_BASE):
    def __init__(self):
        super().__init__()                                        # env.py:17
        self.action_space = reference_action_space()              # env.py:19
        self.observation_space = reference_observation_space()    # env.py:20-25
        self._gameboard = Board(QEvalClassic())                   # env.py:26
        self._reward_map = {"win": 1.0, "loss": -1.0, "draw": 0.0, "otherwise": 0.0}  # env.py:27-32

    def step(self, action, verbose=False):
        cur_player = self.turn() % 2                              # env.py:35
        try:
            self._gameboard.make_move((action[0], action[1]))     # env.py:36-40
        except Exception as e:                                    # env.py:41-43 noop
            from ._native import QtttNativeError
            if isinstance(e, QtttNativeError):
                raise                                             # a broken device is not a noop
            if verbose:
                print('noop (i.e. invalid) move...', e)
        obs = self._observation()
        p1_round, p2_round = self._gameboard.check_win()          # env.py:48
        r = (-1 ** cur_player) * float(p1_round > 0 or p2_round > 0)   # env.py:49, verbatim precedence
        terminated = (p1_round > 0 or p2_round > 0) or self.turn() > 8  # env.py:51
        return obs, r, terminated, False, {}

    def reset(self, *, seed=None, options=None):
        self.__init__()                                           # env.py:56
        return self._observation(), {}

    def render(self):
        displayBoard(self._gameboard)

    def observ(self):
        return self._observation()

    def turn(self):
        return len(self._gameboard.moves)                         # env.py:65-66

    def _reward(self):
        """env.py:87-112 (a helper `step` does not call): player 1's reward from the full check_win pair — the player whose
        line was completed in the EARLIER round wins, a player with no line counts as round 10; equal rounds = 'otherwise'."""
        p1_round, p2_round = self._gameboard.check_win()
        p1 = 10 if p1_round < 0 else p1_round
        p2 = 10 if p2_round < 0 else p2_round
        if p1 == p2:
            return self._reward_map["otherwise"]
        return self._reward_map["win" if p1 < p2 else "loss"]

    def _observation(self):
        # env.py:68-85.  Pure list bookkeeping over what qttt_export returned.
        board = self._gameboard.board
        on_board = set(board)
        q1, q2 = [], []
        for m in self._gameboard.moves:
            if m[-1] not in on_board:
                (q2 if m[-1] % 2 else q1).append(m[:-1])
        return {"q_states_p1": q1, "q_states_p2": q2, "classical": board, "turn": self.turn() % 2}
