"""Without a usable HIP device the product must fail loudly — there is no CPU fallback. Runs only
where no GPU is visible (the authoring container); skipped on the GPU box."""
import numpy as np
import pytest

from clipper_amd import _abi as abi


def _no_gpu():
    try:
        return abi.device_count() <= 0
    except Exception:
        return True


pytestmark = pytest.mark.skipif(not _no_gpu(), reason="a HIP device is visible")


def test_context_creation_fails_with_a_message():
    with pytest.raises(RuntimeError) as e:
        abi.HipClipper()
    assert "no HIP device" in str(e.value) or "CPU fallback" in str(e.value)


def test_stand_alone_entry_points_fail_with_a_message():
    P = np.random.default_rng(0).random((3, 10))
    with pytest.raises(RuntimeError) as e:
        abi.knn(P, P, 1)
    assert "no HIP device" in str(e.value)
    with pytest.raises(RuntimeError) as e:
        abi.distance_based_correspondences(P, P, 1, 0.1, True)
    assert "no HIP device" in str(e.value)


def test_bad_arguments_are_rejected_before_any_device_work():
    P = np.random.default_rng(0).random((4, 10))      # 4 coordinates: not supported
    with pytest.raises(RuntimeError) as e:
        abi.knn(P, P, 1)
    assert "2 or 3 coordinates" in str(e.value)
    P3 = P[:3]
    with pytest.raises(RuntimeError) as e:
        abi.knn(P3, P3, 17)
    assert "knn must be in 1..16" in str(e.value)
