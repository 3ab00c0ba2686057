"""ctypes binding of ``libslime_hip.so`` (C ABI declared in ``include/slime_hip.h``).

There is deliberately NO fallback: if the HIP library is missing or does not export the ABI the
import of a compute entry point raises.  (``tests/test_host_logic.py::test_library_exports_exactly_the_header``
checks, without a GPU, that the product library exports every symbol the header declares and nothing else.)

Two builds of the same sources exist (slime_amd/csrc/Makefile):
  * ``libslime_hip.so``       the product: the header's entry points, no mutable process state;
  * ``libslime_hip_diag.so``  ``-DSLIME_DIAG``: adds process-global tuning / ablation hooks and the measured-alternative GEMM
                              kernels.  Only ``tools/`` and the tile-forcing tests load it (``load_diag()`` / ``diag()``).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLIME_HIP_LIBRARY: run against another build of the SAME ABI (tools/build_variants.sh A/B libraries); there is still no fallback --
# a missing file raises in load()
LIB_PATH = os.environ.get("SLIME_HIP_LIBRARY") or os.path.join(_HERE, "libslime_hip.so")
DIAG_LIB_PATH = os.path.join(_HERE, "libslime_hip_diag.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "slime_hip.h")
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 7
BF16, F16, F32, U8 = 0, 1, 2, 3
(EPI_BIAS_T, EPI_BIAS_QUICKGELU_T, EPI_BIAS_GELU_T, EPI_BIAS_F32, EPI_BIAS_RESID_F32, EPI_BIAS_RESID_F32_LN, EPI_BIAS_RESID_T,
 EPI_BIAS_GELU_MIX_T, EPI_BIAS_RESID_SPLIT_LN) = range(9)

c_void_p, c_int, c_long, c_float, c_size_t = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t


class VitDesc(C.Structure):
    _fields_ = [("hidden", c_int), ("inter", c_int), ("heads", c_int), ("layers_run", c_int),
                ("image", c_int), ("patch", c_int), ("kpad", c_int), ("dtype", c_int), ("eps", c_float),
                ("patch_w", c_void_p), ("cls", c_void_p), ("pos", c_void_p),
                ("pre_ln_w", c_void_p), ("pre_ln_b", c_void_p),
                ("w_qkv", c_void_p), ("b_qkv", c_void_p), ("colsum_qkv", c_void_p),
                ("w_o", c_void_p), ("b_o", c_void_p),
                ("w_fc1", c_void_p), ("b_fc1", c_void_p), ("colsum_fc1", c_void_p), ("w_fc2", c_void_p), ("b_fc2", c_void_p),
                ("w_qkv_frag", c_void_p), ("w_o_frag", c_void_p), ("w_fc1_frag", c_void_p), ("w_fc2_frag", c_void_p), ("patch_w_frag", c_void_p)]


class GemmArgs(C.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_int), ("B", c_void_p), ("bias", c_void_p), ("C", c_void_p), ("ldc", c_int),
                ("M", c_int), ("N", c_int), ("K", c_int), ("dtype", c_int), ("epilogue", c_int),
                ("ln_stats", c_void_p), ("ln_groups", c_int), ("ln_colsum", c_void_p), ("ln_eps", c_float),
                ("x16", c_void_p), ("ldx", c_int), ("stats_out", c_void_p), ("B_frag", c_void_p), ("resid", c_void_p), ("ldr", c_int),
                ("A2", c_void_p), ("mix_gates", c_void_p), ("lo8", c_void_p), ("ldlo", c_int), ("row_map", c_void_p)]


class ResamplerDesc(C.Structure):
    _fields_ = [("dim", c_int), ("heads", c_int), ("n_query", c_int), ("n_kv", c_int), ("dtype", c_int),
                ("eps", c_float), ("q_proj", c_void_p), ("pos_k", c_void_p),
                ("ln_kv_w", c_void_p), ("ln_kv_b", c_void_p), ("w_k", c_void_p), ("b_k", c_void_p),
                ("w_v", c_void_p), ("b_v", c_void_p), ("w_o", c_void_p), ("b_o", c_void_p),
                ("ln_post_w", c_void_p), ("ln_post_b", c_void_p),
                ("w_k_frag", c_void_p), ("w_v_frag", c_void_p), ("w_o_frag", c_void_p)]


class MlpDesc(C.Structure):
    _fields_ = [("in_dim", c_int), ("hidden", c_int), ("dtype", c_int),
                ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p), ("w1_frag", c_void_p), ("w2_frag", c_void_p)]


class LlamaAttnDesc(C.Structure):
    _fields_ = [("hidden", c_int), ("n_heads", c_int), ("n_kv_heads", c_int), ("head_dim", c_int), ("dtype", c_int),
                ("w_qkv", c_void_p), ("w_o", c_void_p), ("inv_freq", c_void_p), ("w_qkv_frag", c_void_p), ("w_o_frag", c_void_p)]


class Probe(C.Structure):
    _fields_ = [("layer", c_int), ("kernel", c_int), ("start", c_void_p), ("stop", c_void_p)]


_P = C.POINTER
_SIGNATURES = {
    "slime_abi_version": (c_int, []),
    "slime_last_error": (C.c_char_p, []),
    "slime_gemm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "slime_gemm_kernel_name": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, C.c_char_p, c_size_t]),
    "slime_gemm_packed_b_bytes": (c_size_t, [c_int, c_int]),
    "slime_gemm_pack_b": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "slime_layernorm": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "slime_gemm_b_frag_usable": (c_int, [c_int, c_int]),
    "slime_patch_embed_prenorm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "slime_gemm_ex": (c_int, [_P(GemmArgs), c_void_p]),
    "slime_attention": (c_int, [c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_void_p, c_long, c_long,
                                c_void_p, c_long, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "slime_gate_mix": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "slime_gather_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "slime_gather_rows_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "slime_merge_rows": (c_int, [c_void_p, c_void_p, c_int, c_long, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "slime_resample_ksize": (c_int, [c_int, c_int]),
    "slime_resample_coeffs": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "slime_resize_bicubic_u8": (c_int, [c_void_p, c_int, c_int, c_long, c_void_p, c_long, c_int, c_int, c_void_p, c_void_p,
                                        c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "slime_merge_rows_batched": (c_int, [c_void_p, c_long, c_void_p, c_int, c_long, c_long, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_void_p]),
    "slime_gate_weights": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "slime_gate_premix": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "slime_gate_mix_ex": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_long,
                                  c_long, c_void_p]),
    "slime_select_crops": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "slime_resize_bicubic_u8_batched": (c_int, [c_void_p, c_int, c_long, c_int, c_int, c_long, c_void_p, c_long, c_long, c_int, c_int,
                                                c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "slime_tile_normalize_batched": (c_int, [c_void_p, c_int, c_long, c_int, c_int, c_int, _P(c_float), _P(c_float), c_void_p, c_long,
                                             c_int, c_void_p]),
    "slime_tile_normalize": (c_int, [c_void_p, c_int, c_int, c_int, _P(c_float), _P(c_float), c_void_p, c_int, c_void_p]),
    "slime_router_scores": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "slime_router_select": (c_int, [c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "slime_router_batched_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "slime_router_scores_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                            c_void_p, c_void_p]),
    "slime_router_select_batched": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "slime_vit_workspace_bytes": (c_size_t, [_P(VitDesc), c_int]),
    "slime_vit_residual_epilogue": (c_int, []),
    "slime_vit_check": (c_int, [_P(VitDesc)]),
    "slime_mfma_stream_probe": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, c_size_t, _P(C.c_double), c_void_p]),
    "slime_vit_forward": (c_int, [_P(VitDesc), c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "slime_vit_forward_ex": (c_int, [_P(VitDesc), c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                     c_size_t, c_void_p, _P(Probe)]),
    "slime_vit_forward_states": (c_int, [_P(VitDesc), c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "slime_resampler_workspace_bytes": (c_size_t, [_P(ResamplerDesc), c_int]),
    "slime_resampler_forward": (c_int, [_P(ResamplerDesc), c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    "slime_mlp_workspace_bytes": (c_size_t, [_P(MlpDesc), c_int]),
    "slime_mlp_forward": (c_int, [_P(MlpDesc), c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "slime_gated_workspace_bytes": (c_size_t, [_P(MlpDesc), _P(ResamplerDesc), c_int]),
    "slime_gated_forward": (c_int, [_P(MlpDesc), _P(ResamplerDesc), c_void_p, c_int, c_void_p, c_int, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "slime_adapter_workspace_bytes": (c_size_t, [_P(MlpDesc), _P(ResamplerDesc), _P(ResamplerDesc), c_int, c_int]),
    "slime_adapter_forward": (c_int, [_P(MlpDesc), _P(ResamplerDesc), c_void_p, c_int, _P(ResamplerDesc), c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p, c_int, c_long, c_void_p, c_size_t, c_void_p]),
    "slime_splice_rows": (c_int, [c_void_p, c_int, c_long, c_void_p, c_int, c_long, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "slime_rope": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_float, c_int, c_void_p]),
    "slime_prefill_attention": (c_int, [c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_void_p, c_long,
                                        c_long, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "slime_llama_attn_workspace_bytes": (c_size_t, [_P(LlamaAttnDesc), c_int, c_int]),
    "slime_llama_attn_forward": (c_int, [_P(LlamaAttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                         c_void_p, c_size_t, c_void_p]),
    "slime_llama_attn_forward_resid": (c_int, [_P(LlamaAttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                               c_void_p, c_size_t, c_void_p]),
}

# diagnostic build only (libslime_hip_diag.so): process-global hooks, never exported by the product library
_DIAG_SIGNATURES = {
    "slime_gemm_force_tile": (None, [c_int]),
    "slime_gemm_set_sched": (None, [c_int]),
    "slime_gemm_set_ablation": (None, [c_int]),
    "slime_gemm_set_group_m": (None, [c_int]),
    "slime_gemm_set_db_ablation": (None, [c_int]),
    "slime_vit_set_skip_mask": (None, [c_int]),
    "slime_gemm_set_shape_tile": (None, [c_int, c_int, c_int]),
    "slime_gemm_set_debug": (None, [c_void_p]),
    "slime_attention_set_debug": (None, [c_void_p]),
    "slime_attention_set_variant": (None, [c_int]),
    "slime_prefill_set_variant": (None, [c_int]),
    "slime_prefill_set_debug": (None, [c_void_p]),
    "slime_attention_set_ablation": (None, [c_int]),
}

_lib = None
_product = None
_diag = None


class SlimeHipError(RuntimeError):
    pass


def header_symbols():
    """Function names declared in include/slime_hip.h (used by the ABI test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slime_[a-z0-9_]+)\s*\(", text)))


def csrc_digest() -> str:
    """sha1 over the kernel sources (csrc/*.hip, *.h, *.inc, Makefile + include/slime_hip.h) in name order: identifies the BUILD a
    profile was taken on (profiles/*_pmc_kernels.json: _meta.csrc_sha) independently of the commit that happens to hold it."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) +
                   [os.path.join(CSRC, "Makefile"), HEADER_PATH])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build(verbose: bool = False) -> str:
    """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, f"-j{max(4, min(16, os.cpu_count() or 4))}"]      # product + diagnostic library
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise SlimeHipError("building libslime_hip.so failed")
    return LIB_PATH


def _bind(path: str, signatures) -> C.CDLL:
    lib = C.CDLL(path)
    for name, (res, args) in signatures.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    got = lib.slime_abi_version()
    if got != ABI_VERSION:
        raise SlimeHipError(f"{os.path.basename(path)} ABI version {got}, expected {ABI_VERSION}")
    return lib


def load():
    """The library every wrapper calls: the product library, unless a diagnostic session swapped in
    ``libslime_hip_diag.so`` (``load_diag()`` / ``diag()``).  Raises if it is missing (no CPU fallback exists)."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlimeHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C slime_amd/csrc`.  slime_amd has no CPU fallback.")
    _product = _bind(LIB_PATH, _SIGNATURES)
    _lib = _product
    return _lib


def load_diag():
    """Switch this process to the diagnostic build (tools/): same ABI plus the ``slime_*_set_*`` / ``force_tile`` hooks."""
    global _lib, _diag
    if _diag is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise SlimeHipError(f"{DIAG_LIB_PATH} not found: `make -C slime_amd/csrc` builds it next to the product library")
        _diag = _bind(DIAG_LIB_PATH, {**_SIGNATURES, **_DIAG_SIGNATURES})
    _lib = _diag
    return _lib


class diag:
    """``with _lib.diag() as lib:`` -- run the enclosed calls on the diagnostic build, then switch back to the product."""

    def __enter__(self):
        load()
        return load_diag()

    def __exit__(self, *exc):
        global _lib
        _lib = _product
        return False


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().slime_last_error().decode("utf-8", "replace")
        raise SlimeHipError(f"{what or 'libslime_hip'} failed (code {rc}): {msg}")
