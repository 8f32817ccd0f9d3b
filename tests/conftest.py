import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test collected on a box without a GPU: run with -m 'not gpu' here")
    # the native library is the product: its absence is a failure, never a skip
    import ace355
    from ace355 import native
    native.lib()
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def full_dit_seed4(gpu_device):
    """The real architecture (24 layers, 2048 hidden, 1.575 B parameters) with weightgen seed-4 "test" weights, i.e. the
    weights the full-size fixtures G4 / G11 / G12 / G13 were captured with; 6.3 GB of fp32 streamed tensor by tensor through
    the C ABI once per session.  Yields (dit, cfg, null_condition_emb, weight_checksum)."""
    import torch
    import ace355
    from ace355 import native, weightgen
    from ace355.dit import NativeDit
    cfg = ace355.DitConfig()
    dit = NativeDit(cfg, gpu_device)
    wsum = 0.0
    for name, shape in cfg.weight_shapes().items():
        wt = weightgen.make_dit_weights({name: shape}, cfg.hidden_size, seed=4, mode="test")[name]
        wsum += float(wt.double().abs().sum())
        native.check(dit._lib.ace355_dit_load_tensor(dit._h, name.encode(), native.ptr(wt.contiguous()), 0, wt.numel(), 0), name)
    native.check(dit._lib.ace355_dit_finalize(dit._h), "finalize")
    null = weightgen.make_null_condition_emb(cfg.hidden_size, seed=4)
    yield dit, cfg, null, wsum
    dit.close()
