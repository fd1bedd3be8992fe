"""huffman_amd -- MI355X (gfx950) Huffman codec behind the module boundaries of the
liuyuxivapor/huffman Chisel encoder (stats, sort, entropy, tree, adjust, codebook, encode,
decode).  The compute path is libhuffhip.so (hand-written HIP); see include/huffhip.h.
"""
from ._lib import (HF_CHUNK_SYMS, HF_CODEBOOK_DEV_BYTES, HF_LAYOUT_BF16_EXP, HF_LAYOUT_FP8,
                   HF_LAYOUT_BITPLANE8, HF_LAYOUT_FP4_E2M1_EXP, HF_LAYOUT_FP8_E4M3_EXP, HF_LAYOUT_FP16_EXP, HF_LAYOUT_U8, HF_MODE_AUTO, HF_MODE_BEST, HF_MODE_BYPASS, HF_MODE_HUFFMAN, LIB_PATH,
                   HuffhipError, lib)

__all__ = [
    "HF_CHUNK_SYMS", "HF_CODEBOOK_DEV_BYTES", "HF_LAYOUT_BF16_EXP", "HF_LAYOUT_FP8",
    "HF_LAYOUT_BITPLANE8", "HF_LAYOUT_FP4_E2M1_EXP", "HF_LAYOUT_FP8_E4M3_EXP", "HF_LAYOUT_FP16_EXP", "HF_LAYOUT_U8", "HF_MODE_AUTO", "HF_MODE_BEST", "HF_MODE_BYPASS", "HF_MODE_HUFFMAN", "LIB_PATH",
    "HuffhipError", "lib",
]


def __getattr__(name):
    # the module mirrors need torch; import them lazily so `import huffman_amd` stays cheap
    if name in ("Context", "SymbolStat", "SymbolSort", "ShannonEntropy", "HuffmanTreeBuilder",
                "tree_adjust", "EntropyAwareTreeBuilder", "TreeQualityAnalyzer", "MemWrapper",
                "DeviceCodebook", "EncodedStream", "EntropyAwareHuffmanSystem",
                "ParrallelHuffmanDecoder", "default_context", "layout_of"):
        from . import api
        return getattr(api, name)
    raise AttributeError(name)
