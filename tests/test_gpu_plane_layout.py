"""BatchedFft.alloc_planes: the four planes of a transform carved out of one allocation with the layout that keeps the
two inputs in one 32-GiB region of the card's address space and gives each output a region of its own.
The layout is a performance matter; here: the planes lie where the docstring says, do not overlap, keep the arena
alive, and a transform through them matches the oracle like any other planes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GIB = 1 << 30


@pytest.fixture(autouse=True)
def _give_the_arena_back():
    """An 80-GiB allocation must not stay in torch's cache behind the tests that follow (the library's own hipMalloc
    calls do not make torch release cached blocks)."""
    yield
    import gc
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float64, 1e-12)])
def test_alloc_planes_layout_and_parity(oracle_mod, dtype, tol):
    from pragma_dsp_amd.batch import BatchedFft
    n, rows = 4096, 16384 if dtype == torch.float32 else 8192   # 256-MiB planes: the smallest that get the layout
    plan = BatchedFft(n, "cuda:0", dtype=dtype)
    small = plan.alloc_planes(64)                                 # small planes: plain allocations
    assert plan.arena is None and len({t.data_ptr() for t in small}) == 4
    del small
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
    re.normal_()
    im.normal_()
    plan.forward(re, im, out=(ore, oim))
    torch.cuda.synchronize()
    sel = torch.tensor([0, 1, rows // 2, rows - 1], device=re.device)   # first, middle and last rows of the planes
    wre, wim = oracle_mod.Plan(n).forward_complex(re[sel].cpu().numpy().astype(np.float64), im[sel].cpu().numpy().astype(np.float64))
    got = ore[sel].cpu().numpy().astype(np.float64) + 1j * oim[sel].cpu().numpy().astype(np.float64)
    want = wre + 1j * wim
    assert (np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)).max() <= tol
    # real input: no imaginary input plane
    r2, none, o2, p2 = plan.alloc_planes(64, real_input=True)
    assert none is None and r2.shape == o2.shape == p2.shape == (64, n)
    # the planes keep their allocation alive after the plan has let go of it
    keep = ore
    plan.arena = None
    del re, im, oim, r2, o2, p2
    torch.cuda.empty_cache()
    assert torch.isfinite(keep).all()


class _DevView:
    """A raw device pointer as a torch tensor (no copy) through __cuda_array_interface__."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


@pytest.mark.parametrize("scalar_bytes,tol", [(4, 1e-5), (8, 1e-12)])
def test_c_abi_planes_alloc_layout_and_parity(oracle_mod, scalar_bytes, tol):
    """pdsp_planes_alloc / pdsp_planes_free: the same layout through the C ABI (what a C or JS caller of the
    device-pointer family uses), a transform through the returned planes against the oracle, argument errors."""
    import ctypes as C
    from pragma_dsp_amd import _capi
    lib, vp = _capi.lib, C.c_void_p
    n, rows = 4096, (16384 if scalar_bytes == 4 else 8192)   # 256-MiB planes: the smallest that get the layout
    plan = vp()
    _capi.check(lib.pdsp_plan_create(n, -1, C.byref(plan)))
    try:
        ptrs = [vp() for _ in range(4)]
        arena, nbytes = vp(), C.c_ulonglong(0)
        _capi.check(lib.pdsp_planes_alloc(plan, rows, scalar_bytes, 0, *[C.byref(p) for p in ptrs], C.byref(arena), C.byref(nbytes)))
        plane = rows * n * scalar_bytes
        base = ptrs[0].value
        if nbytes.value:  # the layout: one allocation of 80 GiB + a plane
            assert nbytes.value == 80 * GIB + plane
            assert ptrs[1].value == base + ((plane + 255) & ~255)
            assert ptrs[2].value == base + 40 * GIB and ptrs[3].value == base + 80 * GIB
        else:
            assert len({p.value for p in ptrs}) == 4
        dt, ts = (torch.float32, "<f4") if scalar_bytes == 4 else (torch.float64, "<f8")
        re, im, ore, oim = (torch.as_tensor(_DevView(p.value, (rows, n), ts), device="cuda:0") for p in ptrs)
        re.normal_()
        im.normal_()
        torch.cuda.synchronize()
        fwd = lib.pdsp_fft_forward_complex_f32 if scalar_bytes == 4 else lib.pdsp_fft_forward_complex_f64
        _capi.check(fwd(plan, rows, ptrs[0], ptrs[1], ptrs[2], ptrs[3], None))
        torch.cuda.synchronize()
        sel = torch.tensor([0, 1, rows // 2, rows - 1], device=re.device)
        wre, wim = oracle_mod.Plan(n).forward_complex(re[sel].cpu().numpy().astype(np.float64), im[sel].cpu().numpy().astype(np.float64))
        got = ore[sel].cpu().numpy().astype(np.float64) + 1j * oim[sel].cpu().numpy().astype(np.float64)
        want = wre + 1j * wim
        assert (np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)).max() <= tol
        del re, im, ore, oim
        assert lib.pdsp_planes_free(arena) == 0
        # real input: no imaginary input plane
        small = C.c_ulonglong(1)
        _capi.check(lib.pdsp_planes_alloc(plan, 32, scalar_bytes, 1, *[C.byref(p) for p in ptrs], C.byref(arena), C.byref(small)))
        assert ptrs[1].value is None and ptrs[0].value and ptrs[2].value and ptrs[3].value
        assert small.value == 0   # small planes: plain allocations
        assert lib.pdsp_planes_free(arena) == 0 and lib.pdsp_planes_free(None) == 0
        # argument errors
        assert lib.pdsp_planes_alloc(plan, 0, 4, 0, *[C.byref(p) for p in ptrs], C.byref(arena), None) == _capi.ERR_BAD_ARG
        assert lib.pdsp_planes_alloc(plan, rows, 2, 0, *[C.byref(p) for p in ptrs], C.byref(arena), None) == _capi.ERR_BAD_ARG
        assert lib.pdsp_planes_alloc(None, rows, 4, 0, *[C.byref(p) for p in ptrs], C.byref(arena), None) == _capi.ERR_BAD_ARG
    finally:
        lib.pdsp_plan_destroy(plan)
