import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The whole-step oracle tests at the quoted batch sizes: their oracle evaluations run in worker processes beside the rest
# of the suite (tests/oracle_pool.py), so they are moved to the END of the run (stable order otherwise).  The small
# oracle cases (B <= 64 on the toy nets) still run first, so that `-x` reaches an oracle comparison before the HIP-vs-HIP checks.
_POOLED = ("test_full_arch_stage1_step_at_batch_256_against_the_oracle", "test_full_arch_stage2_frozen_decoder_at_batch_256",
           "test_128px_six_level_arch_at_its_per_gpu_batch_of_64", "test_full_arch_at_the_quoted_batch_of_256_stamps",
           "test_full_arch_64_stamps", "test_deep_arch_128px_at_its_per_gpu_batch_of_64",
           "test_ten_bands_on_the_reference_architecture_bf16")


def pytest_collection_modifyitems(config, items):
    tail = [it for it in items if it.name.split("[")[0] in _POOLED]
    if tail:
        items[:] = [it for it in items if it.name.split("[")[0] not in _POOLED] + tail


@pytest.fixture(scope="session", autouse=True)
def _oracle_workers(request):
    """Starts the oracle worker pool before the first test of a GPU session - i.e. before the first GPU call - when one
    of the pooled tests was selected; CPU sessions never start it."""
    from tests import oracle_pool

    wanted = [it for it in request.session.items if it.name.split("[")[0] in _POOLED]
    if wanted and os.environ.get("DV_ORACLE_POOL", "1") != "0":
        oracle_pool.start()
    yield
    if wanted:
        st = oracle_pool.stats()
        print(f"\n[oracle pool] fetched {st['fetched']} evaluations (waited {st['waited_s']:.1f} s for them), "
              f"{st['inline']} computed inline")
    oracle_pool.stop()
