"""Multi-rank orchestration of the sharded index on CPU: world_size 2 and 4 with
the gloo backend, device operations replaced by numpy + oracle stand-ins
(tests/sharded_worker.py).  Checks geometry (local region split), the padded
slot format, canonical (source rank, index) insert order and result routing
against one aggregate oracle table."""
import pytest

from tests.sharded_check import run_and_check


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("scheme", [0, 1])
def test_sharded_orchestration_gloo_cpu(world, scheme):
    run_and_check(world, "cpu", mem_p_local=14, scheme=scheme, seed=3 + world)


def test_sharded_eight_way_small():
    run_and_check(8, "cpu", mem_p_local=12, scheme=0, seed=11)
