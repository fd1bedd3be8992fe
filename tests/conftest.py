import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def hf():
    """The HIP-backed host API; fails loudly when libhuffhip.so is absent."""
    import huffman_amd
    huffman_amd.lib()
    from huffman_amd import api
    return api


def gaussian_bf16(n: int, seed: int = 1234, std: float = 0.02) -> np.ndarray:
    """bf16 bit patterns (uint16) of N(0, std^2) fp32 values rounded to nearest even."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n).astype(np.float32) * np.float32(std)).view(np.uint32)
    r = ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)
    return r


def skewed_symbols(n: int, seed: int, kind: str) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == "geometric":
        return np.minimum(rng.geometric(0.5, n) - 1, 255).astype(np.uint8)
    if kind == "one":
        return np.full(n, 77, np.uint8)
    if kind == "two":
        return np.where(rng.random(n) < 0.999, 5, 200).astype(np.uint8)
    if kind == "fib":
        fib = [1, 1]
        while len(fib) < 40:
            fib.append(fib[-1] + fib[-2])
        p = np.array(fib[::-1], np.float64)
        p /= p.sum()
        return rng.choice(40, n, p=p).astype(np.uint8)
    raise ValueError(kind)
