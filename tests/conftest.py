"""pytest configuration: the ``gpu`` marker and the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: soak-style repetition (more fuzz seeds of a case the suite already holds); "
                                       "skipped unless RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="soak-style repetition: set RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def seeds(fast, slow):
    """parametrize values: the `fast` seeds always run, the `slow` ones only with RUN_SLOW=1."""
    return list(fast) + [pytest.param(s_, marks=pytest.mark.slow) for s_ in slow]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
