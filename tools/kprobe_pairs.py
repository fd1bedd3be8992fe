"""Time the beat kernel (hf_encode_pairs: one {code, length, mode} beat per symbol) alone."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from huffman_amd import api as hf
from huffman_amd import _lib as L

n = 1 << 30
dev = "cuda:0"
w = (torch.randn(n, device=dev) * 0.02).to(torch.bfloat16)
b8 = torch.randint(0, 256, (n,), device=dev, dtype=torch.uint8)
s = hf.EntropyAwareHuffmanSystem()
beats = torch.empty(n, dtype=torch.int32, device=dev)
for name, t, bytes_in in (("bf16", w, 2), ("u8", b8, 1)):
    cb = s.codebook(s.histogram(t), hf.HF_MODE_HUFFMAN)
    lay = hf.layout_of(t)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(4):
        if it == 1:
            ev[0].record()
        hf.check(L.lib().hf_encode_pairs(s.ctx.handle, C.c_void_p(t.data_ptr()), n, lay,
                                         C.c_void_p(cb.ptr), C.c_void_p(beats.data_ptr())), "pairs")
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 3
    print(f"{name}: {ms:.3f} ms per Gi symbols, {(bytes_in + 4) * n / ms / 1e6:.0f} GB/s moved")
