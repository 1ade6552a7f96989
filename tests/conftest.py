import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "terra-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


# ImageNet VGG16 weights cannot be fetched offline: the tests opt in to the deterministic stand-in trunk (same architecture and
# FLOPs; the golden fixtures were generated with the same stand-in, tests/golden/make_golden.py)
os.environ.setdefault("TERRAGAN_ALLOW_STANDIN_VGG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
