"""The 36-action indexing L3 callers use (mcts.py:339-350): action a <-> the unordered pair of squares
(i, j), i < j, in lexicographic order (0,1), (0,2) … (0,8), (1,2) … (7,8).  `Env.step` takes the pair
(env.py:37-38), `MCTS` and `VecEnv.expand` the index; these helpers convert between the two, for one
action on the host or for a whole batch on the device (no host sync)."""
import torch

_PAIRS = tuple((i, j) for i in range(9) for j in range(i + 1, 9))
_INDEX = {p: a for a, p in enumerate(_PAIRS)}
N_ACTIONS = 36


def ind2move(n):
    """mcts.py:339-343: action index -> (i, j), i < j."""
    return _PAIRS[n]


def move2ind(i, j):
    """mcts.py:345-350: (i, j) in either order -> action index."""
    return _INDEX[(i, j) if i < j else (j, i)]


_tables = {}


def _device_tables(device):
    t = _tables.get(device)
    if t is None:
        pairs = torch.tensor(_PAIRS, dtype=torch.uint8, device=device)                # [36, 2]
        index = torch.full((9, 9), 255, dtype=torch.uint8, device=device)             # 255 = not an action (i == j)
        for a, (i, j) in enumerate(_PAIRS):
            index[i, j] = index[j, i] = a
        t = _tables[device] = (pairs, index)
    return t


def action36_to_pairs(action36):
    """u8/int tensor [...] of action indices -> u8 [..., 2] of squares (what VecEnv.step takes).
    Indices outside 0..35 become (255, 255), which the step treats as a noop (env.py:41)."""
    pairs, _ = _device_tables(action36.device)
    a = action36.to(torch.int64)
    ok = (a >= 0) & (a < N_ACTIONS)
    out = pairs[a.clamp(0, N_ACTIONS - 1)]
    return torch.where(ok.unsqueeze(-1), out, torch.full_like(out, 255))


def pairs_to_action36(pairs):
    """u8 tensor [..., 2] of squares (either order) -> u8 [...] of action indices; 255 where the pair is not an
    action (same square twice, or a square outside 0..8)."""
    _, index = _device_tables(pairs.device)
    p = pairs.to(torch.int64)
    ok = ((p >= 0) & (p < 9)).all(dim=-1)
    q = p.clamp(0, 8)
    out = index[q[..., 0], q[..., 1]]
    return torch.where(ok, out, torch.full_like(out, 255))


def legal_mask_to_bool(legal):
    """int64 [...] legal-action bit masks (VecEnv.node_info / expand: bit a = action a is legal, mcts.py:20-27)
    -> bool [..., 36] (the shape of GameState.action_mask, mcts.py:87-91)."""
    bits = torch.arange(N_ACTIONS, dtype=torch.int64, device=legal.device)
    return ((legal.unsqueeze(-1) >> bits) & 1).to(torch.bool)
