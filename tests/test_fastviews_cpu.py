"""qtttgym_amd/_fastviews.so (csrc/fastviews.cpp): the host helper of the default VecEnv.step() — one allocation and the
eight tensors a step returns as views of it.  No GPU: the same carve on CPU tensors against vec_env.py's own Python
construction (dtype, shape, strides, offset of every view, one shared storage), the layout arithmetic at the edges, and
that the views behave like ordinary tensors (write through, arithmetic, sub-views, freed with their last owner)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _module():
    import __graft_entry__ as g
    g.build_fastviews()
    from qtttgym_amd import _fastviews
    return _fastviews


def test_carve_equals_the_python_construction():
    fv = _module()
    from qtttgym_amd import vec_env
    cpu = torch.device("cpu")
    for n in (0, 1, 2, 63, 64, 65, 511, 512, 513, 1000, 4096, 100003, 1 << 20):
        lay = vec_env._layout(n)
        assert tuple(fv.offsets(n)) == lay and all(o % 512 == 0 for o in lay)
        a, base_a = vec_env._carve_py(n, cpu)
        b, base_b = fv.carve(n, cpu)
        assert len(a) == len(b) == 8
        for k, (x, y) in enumerate(zip(a, b)):
            assert x.dtype == y.dtype and x.shape == y.shape and x.stride() == y.stride() and y.is_contiguous(), (n, k)
            assert x.data_ptr() - base_a == y.data_ptr() - base_b == (lay[k] if n else y.data_ptr() - base_b), (n, k)
            assert y.device == cpu and not y.requires_grad
        if n:
            st = {t.untyped_storage().data_ptr() for t in b}
            assert len(st) == 1 and b[0].untyped_storage().nbytes() == lay[8]
    assert [t.dtype for t in fv.carve(5, cpu)[0]] == [torch.float32, torch.bool, torch.uint8, torch.uint8, torch.uint8,
                                                      torch.uint8, torch.int8, torch.uint8]


def test_views_are_ordinary_tensors():
    fv = _module()
    t, base = fv.carve(1000, torch.device("cpu"))
    reward, term, q1, l1, q2, l2, classical, turn = t
    for x in t:
        x.zero_()
    q1[3, 1, 0] = 7
    classical[5:9] = -1
    reward[:] = -0.0
    assert q1.sum().item() == 7 and classical.sum().item() == -36 and term.any().item() is False
    assert reward.view(torch.int32)[0].item() == -(2 ** 31)                    # the sign bit survives (env.py:49)
    assert (reward + 1)[:2].tolist() == [1.0, 1.0] and classical[5:9].shape == (4, 9)
    # no view overlaps another: writing each one fully leaves the ones behind it alone
    for x in t:
        x.zero_()
    for k, x in enumerate(t):
        x.fill_(True if x.dtype == torch.bool else 1)
        for j, y in enumerate(t):
            if j > k:
                assert not y.to(torch.float32).any().item(), (k, j)
    # the allocation lives exactly as long as its last view
    keep = classical[0]
    ptr = keep.untyped_storage().data_ptr()
    del t, reward, term, q1, l1, q2, l2, classical, turn, x, y
    assert keep.untyped_storage().data_ptr() == ptr and keep.tolist() == [1] * 9


def test_vec_env_uses_it_when_built():
    _module()
    import importlib
    from qtttgym_amd import vec_env
    importlib.reload(vec_env)
    assert vec_env._fastviews is not None and vec_env._carve is vec_env._fastviews.carve
