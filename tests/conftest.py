import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import json

    def load(name):
        with open(os.path.join(REPO, 'tests', 'golden', name)) as f:
            return json.load(f)
    return load


@pytest.fixture(scope='session', autouse=True)
def _build_oracle():
    import oracle
    oracle.build()
