"""The sharded path with the real HIP kernels: 2 and 4 ranks share the one GPU of
the box (collective = gloo staged through host memory; RCCL needs one GPU per
rank), checked against one aggregate oracle table."""
import pytest

from tests.sharded_check import run_and_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,scheme", [(2, 0), (2, 1), (4, 0)])
def test_sharded_hip_ranks_on_one_gpu(world, scheme):
    run_and_check(world, "gpu", mem_p_local=16, scheme=scheme, seed=20 + world, timeout=900)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_pipelined_two_lanes(world):
    """run_pipelined (two batches in flight, exchanges overlapping local kernels) must
    give exactly the results of the batches run one after the other."""
    run_and_check(world, "gpu-pipe", mem_p_local=16, scheme=0, seed=30 + world, timeout=900)
