"""CPU suite: the C-ABI library loads and exports what include/huffhip.h declares; the
host-only entry points compute; nothing needs a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "huffhip.h")


@pytest.fixture(scope="module")
def lib():
    import huffman_amd
    return huffman_amd.lib()


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(lib):
    from huffman_amd import _lib
    names = declared_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in huffhip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_header_constants_match_binding():
    from huffman_amd import _lib
    src = open(HEADER).read()
    for name in ("HF_ABI_VERSION", "HF_CHUNK_SYMS", "HF_MAX_CODE_LEN", "HF_CODEBOOK_DEV_BYTES",
                 "HF_LAYOUT_U8", "HF_LAYOUT_BF16_EXP", "HF_LAYOUT_FP8", "HF_LAYOUT_FP16_EXP",
                 "HF_LAYOUT_FP8_E4M3_EXP", "HF_LAYOUT_FP4_E2M1_EXP", "HF_LAYOUT_BITPLANE8",
                 "HF_BITPLANE_BLOCK", "HF_MODE_BYPASS",
                 "HF_MODE_HUFFMAN", "HF_MODE_AUTO"):
        m = re.search(rf"#define\s+{name}\s+\(?(-?\d+)u?\)?", src)
        assert m, name
        assert int(m.group(1)) == getattr(_lib, name), name
    for name in ("HF_ERR_ARG", "HF_ERR_ALIGN", "HF_ERR_RANGE", "HF_ERR_HIP", "HF_ERR_OVERFLOW",
                 "HF_ERR_TIMEOUT", "HF_ERR_CORRUPT", "HF_ERR_NOMEM", "HF_ERR_RCCL"):
        m = re.search(rf"#define\s+{name}\s+\((-?\d+)\)", src)
        assert m and int(m.group(1)) == getattr(_lib, name), name


def test_host_only_entry_points(lib, orc):
    assert lib.hf_abi_version() == 3
    assert lib.hf_error_string(0) == b"ok" and b"aligned" in lib.hf_error_string(-2)
    for n in (0, 1, 16384, 16385, 10**9):
        assert lib.hf_num_chunks(n) == orc.num_chunks(n)
        assert lib.hf_payload_bound_words(n) == orc.payload_bound_words(n)


def test_quality_matches_oracle(lib, orc):
    rng = np.random.default_rng(4)
    for W in (32, 16, 64):
        f = rng.integers(0, 3000, 256).astype(np.uint64)
        _, lens = orc.rp_system_codebook(f, 64, 0, 16)
        a, e, r = C.c_uint16(0), C.c_uint16(0), C.c_uint16(0)
        rc = lib.hf_quality(None, f.ctypes.data_as(C.POINTER(C.c_uint64)),
                            lens.ctypes.data_as(C.POINTER(C.c_uint8)), 256, W, C.byref(a),
                            C.byref(e), C.byref(r))
        assert rc == 0
        assert (a.value, e.value, r.value) == orc.aw_quality(f, lens, W)
    z = np.zeros(256, np.uint64)
    rc = lib.hf_quality(None, z.ctypes.data_as(C.POINTER(C.c_uint64)),
                        lens.ctypes.data_as(C.POINTER(C.c_uint8)), 256, 32, C.byref(a), C.byref(e),
                        C.byref(r))
    assert rc == -3  # the RTL divides by zero here


def test_adaptive_threshold_register_sets(lib, orc):
    from huffman_amd import api
    # adaptive_threshold.scala:41-62 at the system's parameters (depth 256, wtWidth 32)
    at = api.AdaptiveThreshold(256, 32)
    assert at(0x7FFF) == dict(encoding_mode=1, compression_threshold=19660, tree_depth_limit=64,
                              min_freq_threshold=1)
    assert at(0x8000) == dict(encoding_mode=1, compression_threshold=39321, tree_depth_limit=128,
                              min_freq_threshold=2)
    assert at(0x60000)["encoding_mode"] == 1
    assert at(0x60001) == dict(encoding_mode=0, compression_threshold=58982, tree_depth_limit=32,
                               min_freq_threshold=1 << 30)
    # every width and depth against the oracle, the mode against the decision the system uses
    for depth in (2, 4, 16, 100, 256):
        for W in range(2, 65):
            for e in (0, 0x7FFF, 0x8000, 0x40000, 0x60000, 0x60001, 0xFFFFFFFF):
                exp = orc.aw_adaptive_config(e, depth, W)
                if exp is None:
                    with pytest.raises(api.HuffhipError):
                        api.AdaptiveThreshold(depth, W)(e)
                    continue
                got = api.AdaptiveThreshold(depth, W)(e)
                assert tuple(got.values()) == exp, (depth, W, e)
                assert got["encoding_mode"] == orc.aw_adaptive_mode(e)
                assert got["min_freq_threshold"] < (1 << W) and got["tree_depth_limit"] < 256
    cfg = api.L.hf_adaptive_config()
    assert lib.hf_adaptive_threshold(0, 1, 32, C.byref(cfg)) == -1
    assert lib.hf_adaptive_threshold(0, 256, 65, C.byref(cfg)) == -1
    assert lib.hf_adaptive_threshold(0, 256, 32, None) == -1


def test_argument_checks_need_no_gpu(lib):
    # null context / null pointers are rejected before any HIP call
    assert lib.hf_flush(None) == -1
    assert lib.hf_sync(None) == -1
    assert lib.hf_symbol_sort(None, None, 256, None, None) == -1
    assert lib.hf_encode(None, None, 0, 0, None, None, None, 0, None, None) == -1
    assert lib.hf_decode(None, None, 0, None, None, 0, 0, None, None, 0) == -1


def test_no_gpu_means_a_clean_error_not_a_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.hf_ctx_create(C.byref(h), 0, None) == -4 and not h.value
    from huffman_amd import api
    with pytest.raises(RuntimeError):
        api.Context()


def test_product_sources_never_touch_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "huffman_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\boracle\b|liboracle|hf_oracle", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_scala_binding_names_every_entry_point():
    # the JNA interface a maintainer of the reference would add (source only: no JVM here) must
    # declare exactly the functions the header declares
    import os
    import re
    from huffman_amd import _lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "bindings", "scala", "huffman", "hip", "HuffHip.scala")).read()
    scala = set(re.findall(r"def (hf_[a-z0-9_]+)\(", text))
    assert scala == set(L.SIGNATURES)
    # one software twin per module of the reference, under the reference's own class name
    # (symbol_stat.scala:6, symbol_sort.scala:82, shannon_entropy.scala:11,
    # adaptive_threshold.scala:11, tree_build.scala:17, tree_adjust.scala:6,
    # entropy_aware_tree_builder.scala:6, mem_wrapper.scala:6, tree_q_analyser.scala:10,
    # top_encoder.scala:6, decoder/top_decoder.scala:6)
    twins = set(re.findall(r"^class ([A-Za-z_]+)\(", text, flags=re.M))
    modules = {"SymbolStat", "SymbolSort", "ShannonEntropy", "AdaptiveThreshold",
               "HuffmanTreeBuilder", "tree_adjust", "EntropyAwareTreeBuilder", "MemWrapper",
               "TreeQualityAnalyzer", "EntropyAwareHuffmanSystem", "ParrallelHuffmanDecoder"}
    assert modules <= twins, modules - twins
    # the same names exist on the Python side of the boundary
    from huffman_amd import api
    assert all(hasattr(api, m) for m in modules)
    # and the document points at this file instead of repeating it
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert "trait HuffHip extends Library {" not in doc and "HuffHip.scala" in doc


def test_scala_binding_constants_match_the_header():
    hdr = open(HEADER).read()
    scala = open(os.path.join(ROOT, "bindings", "scala", "huffman", "hip", "HuffHip.scala")).read()
    for obj, prefix in (("Layout", "HF_LAYOUT_"), ("Mode", "HF_MODE_"), ("Reserve", "HF_RESERVE_")):
        body = re.search(rf"object {obj} \{{(.*?)\n\}}", scala, re.S)
        if body is None:
            body = re.search(rf"object {obj} \{{(.*?)\}}", scala, re.S)
        pairs = dict((k, int(v)) for k, v in re.findall(r"val (\w+) = (\d+)", body.group(1)))
        assert pairs, obj
        for name, value in pairs.items():
            m = re.search(rf"#define\s+{prefix}{name}\s+(\d+)u?\b", hdr)
            assert m and int(m.group(1)) == value, (obj, name)
        assert len(pairs) == len(re.findall(rf"#define\s+{prefix}\w+\s+\d+u?\b", hdr)), obj


# ---- the Scala binding's signatures against the ctypes ones ------------------------------------
# JNA type of a Scala parameter -> the class of C type it marshals to
_SCALA_CLASS = {
    "Pointer": "ptr", "Long": "i64", "Int": "i32", "String": "str",
    "Array[Long]": "p64", "LongByReference": "p64", "Array[Short]": "p16",
    "ShortByReference": "p16", "Array[Byte]": "p8", "IntByReference": "p32",
    "FloatByReference": "pf32", "PointerByReference": "pptr",
    "ContainerInfo": "struct hf_container_info", "EntropyConfig": "struct hf_entropy_config",
}
# raw memory on the Scala side where the C side takes a struct pointer
_SCALA_RAW_STRUCTS = {("hf_adaptive_threshold", 3): "struct hf_adaptive_config"}


def _ctypes_class(t):
    if t is C.c_void_p:
        return "ptr"
    if t is C.c_char_p:
        return "str"
    if t is C.c_uint64:
        return "i64"
    if t in (C.c_int, C.c_uint32, C.c_uint):
        return "i32"
    if hasattr(t, "_type_") and not isinstance(t._type_, str):  # POINTER(x)
        x = t._type_
        if x in (C.c_int, C.c_uint32):
            return "p32"
        if issubclass(x, C.Structure):
            return "struct " + x.__name__
        return {C.c_uint64: "p64", C.c_uint16: "p16", C.c_uint8: "p8", C.c_float: "pf32",
                C.c_void_p: "pptr"}[x]
    raise AssertionError(f"no class for {t}")


def scala_signature_mismatches(text):
    """Every `def hf_*` of trait HuffHip against _lib.SIGNATURES: argument count, the class of
    each argument (pointer / 64-bit / 32-bit / typed host array / struct) and the result."""
    from huffman_amd import _lib as L
    trait = re.search(r"trait HuffHip extends Library \{(.*?)\n\}", text, re.S).group(1)
    trait = re.sub(r"//[^\n]*", "", trait)
    bad = []
    seen = set()
    for name, args, ret in re.findall(r"def (hf_[a-z0-9_]+)\((.*?)\)\s*:\s*(\w+)", trait, re.S):
        seen.add(name)
        if name not in L.SIGNATURES:
            bad.append(f"{name}: not in the C ABI")
            continue
        res, argtypes = L.SIGNATURES[name]
        params = [a.split(":", 1)[1].strip() for a in args.split(",") if a.strip()]
        if len(params) != len(argtypes):
            bad.append(f"{name}: {len(params)} parameters, the C ABI has {len(argtypes)}")
            continue
        for i, (sc, ct) in enumerate(zip(params, argtypes)):
            want = _ctypes_class(ct)
            got = _SCALA_CLASS.get(sc)
            if got == "ptr" and _SCALA_RAW_STRUCTS.get((name, i)) == want:
                continue
            if got != want:
                bad.append(f"{name}: parameter {i} is {sc} ({got}), the C ABI has {want}")
        if _SCALA_CLASS.get(ret) != _ctypes_class(res):
            bad.append(f"{name}: returns {ret}, the C ABI has {_ctypes_class(res)}")
    for name in sorted(set(L.SIGNATURES) - seen):
        bad.append(f"{name}: missing from trait HuffHip")
    return bad


def test_scala_binding_signatures_match_the_c_abi():
    path = os.path.join(ROOT, "bindings", "scala", "huffman", "hip", "HuffHip.scala")
    text = open(path).read()
    assert scala_signature_mismatches(text) == []
    # the check has teeth: two neighbouring parameters of different classes swapped, a parameter
    # dropped and a 64-bit count declared as Int are all caught
    swapped = text.replace("def hf_symbol_stat_dev(ctx: Pointer, dData: Pointer, n: Long, layout: Int,",
                           "def hf_symbol_stat_dev(ctx: Pointer, dData: Pointer, layout: Int, n: Long,")
    assert swapped != text
    assert any("hf_symbol_stat_dev" in m for m in scala_signature_mismatches(swapped))
    dropped = text.replace("def hf_flush(ctx: Pointer): Int", "def hf_flush(): Int")
    assert any("hf_flush" in m for m in scala_signature_mismatches(dropped))
    narrowed = text.replace("def hf_num_chunks(n: Long): Long", "def hf_num_chunks(n: Int): Long")
    assert any("hf_num_chunks" in m for m in scala_signature_mismatches(narrowed))
