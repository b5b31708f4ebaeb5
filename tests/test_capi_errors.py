"""Error behaviour of the C-ABI, called raw through ctypes (no Python-side validation in between).  The reference routine
has none -- failure is a crash (SURVEY 8b "Errors") --; the replacement returns MPDATA_EINVAL with a text for the calling
thread BEFORE it touches the device, and on a box without a GPU a valid call fails with the HIP error (no CPU fallback:
nothing is computed anywhere else).  Runs without a GPU."""
import ctypes

import pytest


@pytest.fixture(scope="module")
def L(mpdata):
    return mpdata.lib()


def _err(L):
    return L.mpdata_last_error().decode()


@pytest.mark.parametrize("sizes", [(0, 32, 28, 1), (64, 0, 28, 1), (64, 32, 2, 1), (64, 32, 28, 0), (-5, 32, 28, 1)])
def test_bad_sizes_are_refused_before_the_device_is_touched(mpdata, L, sizes):
    p = ctypes.c_void_p()
    rc = L.mpdata_plan_create(ctypes.c_int64(sizes[0]), sizes[1], sizes[2], sizes[3], ctypes.byref(p))
    assert rc == mpdata.EINVAL and "bad sizes" in _err(L) and not p.value
    rc = L.mpdata_advect_scalar2d_device(ctypes.c_int64(sizes[0]), sizes[1], sizes[2], sizes[3],
                                         None, None, None, None, None, None, None, None)
    assert rc == mpdata.EINVAL


def test_null_pointers_and_null_plans(mpdata, L):
    rc = L.mpdata_advect_scalar2d_device(ctypes.c_int64(64), 32, 28, 1, None, None, None, None, None, None, None, None)
    assert rc == mpdata.EINVAL and "null" in _err(L)
    rc = L.mpdata_advect_scalar2d(ctypes.c_int64(64), 32, 28, 1, None, None, None, None, None, None, None)
    assert rc == mpdata.EINVAL
    for fn in (L.mpdata_plan_run, L.mpdata_plan_sync):
        assert fn(None) == mpdata.EINVAL and "null plan" in _err(L)
    assert L.mpdata_plan_create(ctypes.c_int64(64), 32, 28, 1, None) == mpdata.EINVAL


def test_without_a_gpu_a_valid_call_fails_with_the_hip_error(mpdata, L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = ctypes.c_void_p()
    rc = L.mpdata_plan_create(ctypes.c_int64(64), 32, 28, 1, ctypes.byref(p))
    assert rc > 0 and not p.value and "device" in _err(L).lower()      # a hipError_t, not a result from somewhere else
    assert L.mpdata_device_count() <= 0
