import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU runs: bring torch's HIP runtime up before the encoder library's (tests that hand device-resident pictures to
    # the C ABI need both in one process, and torch refuses to initialise after another HIP user has)
    expr = getattr(config.option, "markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
