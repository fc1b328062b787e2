"""BatchedFft.alloc_planes: the four planes of a transform carved out of one allocation with the layout that keeps the
two inputs in one 32-GiB region of the card's address space and gives each output a region of its own.
The layout is a performance matter; here: the planes lie where the docstring says, do not overlap, keep the arena
alive, and a transform through them matches the oracle like any other planes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GIB = 1 << 30


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float64, 1e-12)])
def test_alloc_planes_layout_and_parity(oracle_mod, dtype, tol):
    from pragma_dsp_amd.batch import BatchedFft
    n, rows = 1024, 96
    plan = BatchedFft(n, "cuda:0", dtype=dtype)
    re, im, ore, oim = plan.alloc_planes(rows)
    esize = 4 if dtype == torch.float32 else 8
    plane = rows * n * esize
    for t in (re, im, ore, oim):
        assert t.shape == (rows, n) and t.dtype == dtype and t.is_contiguous()
    free, _total = torch.cuda.mem_get_info()
    if plan.arena is not None:
        base = plan.arena.data_ptr()
        assert re.data_ptr() == base and im.data_ptr() == base + ((plane + 255) & ~255)
        assert ore.data_ptr() == base + 40 * GIB and oim.data_ptr() == base + 80 * GIB
        assert plan.arena.numel() == 80 * GIB + plane
    else:  # a card without 84 GiB free: plain allocations, same contract
        assert len({t.data_ptr() for t in (re, im, ore, oim)}) == 4
    rng = np.random.default_rng(11)
    x, y = rng.standard_normal((rows, n)), rng.standard_normal((rows, n))
    re.copy_(torch.from_numpy(x).to(dtype))
    im.copy_(torch.from_numpy(y).to(dtype))
    plan.forward(re, im, out=(ore, oim))
    torch.cuda.synchronize()
    wre, wim = oracle_mod.Plan(n).forward_complex(re.cpu().numpy().astype(np.float64), im.cpu().numpy().astype(np.float64))
    got = ore.cpu().numpy().astype(np.float64) + 1j * oim.cpu().numpy().astype(np.float64)
    want = wre + 1j * wim
    assert (np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)).max() <= tol
    # real input: no imaginary input plane
    r2, none, o2, p2 = plan.alloc_planes(rows, real_input=True)
    assert none is None and r2.shape == o2.shape == p2.shape == (rows, n)
    # the planes keep their allocation alive after the plan has let go of it
    keep = ore
    plan.arena = None
    del re, im, oim, r2, o2, p2
    torch.cuda.empty_cache()
    assert torch.isfinite(keep).all()
