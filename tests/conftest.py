import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = os.path.join(GOLDEN, "reference_fixtures")
WEIGHTS = os.path.join(FIXTURES, "silero_v31_16k.testtensor")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights_path():
    return WEIGHTS


@pytest.fixture(scope="session")
def weights_blob():
    with open(WEIGHTS, "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def fixture_path():
    return lambda name: os.path.join(FIXTURES, name + ".testtensor")


def run_cli(cmd, data, timeout=120):
    """run a host CLI with `data` on stdin; a run that does not come back fails with what it had written so far (instead of a bare TimeoutExpired)"""
    import subprocess
    env = dict(os.environ, VADC_AMD_TRACE_TEARDOWN="1")      # (stderr is only looked at when the child does not come back)
    try:
        return subprocess.run(cmd, input=data, capture_output=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired as ex:
        raise AssertionError(f"{' '.join(map(str, cmd))} did not finish within {timeout} s; stderr so far: {(ex.stderr or b'').decode(errors='replace')[-1500:]!r}; "
                             f"stdout so far: {len(ex.stdout or b'')} bytes") from None
