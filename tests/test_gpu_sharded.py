"""The sharded path with the real HIP kernels: 2 and 4 ranks share the one GPU of
the box (collective = gloo staged through host memory; RCCL needs one GPU per
rank), checked against one aggregate oracle table."""
import pytest

from tests.sharded_check import run_and_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,scheme", [(2, 0), (2, 1), (4, 0)])
def test_sharded_hip_ranks_on_one_gpu(world, scheme):
    run_and_check(world, "gpu", mem_p_local=16, scheme=scheme, seed=20 + world, timeout=900)
