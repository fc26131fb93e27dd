"""ctypes binding of libsimulst_hip.so (the C ABI declared in include/simulst_hip.h).

The product path has NO fallback: if the shared library is missing or a call
returns a non-zero status, a RuntimeError is raised (the reference's native
precedent surfaces TORCH_CHECK failures the same way,
criterion/best_alignment/best_alignment.cu:233-238 -- but unlike
criterion/best_alignment/__init__.py:18-20 a missing extension is never
silently ignored).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIMULST_LIB_PATH: another build of the same library (a PROBE / DEBUG_HOOKS build under csrc/build_dbg/, tools/ only); the ABI
# version check below applies to it as well
LIB_PATH = os.environ.get("SIMULST_LIB_PATH") or os.path.join(_HERE, "libsimulst_hip.so")

F32, BF16 = 0, 1
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_GLU, EPI_EMF_OUT, EPI_BIAS_F32OUT, EPI_BIAS_RES_GELU = range(7)
ATTN_HARD, ATTN_INFINITE_LOOKBACK, ATTN_WAITK, ATTN_CHUNKWISE = range(4)
(K_LINEAR, K_LAYERNORM, K_EMF_ATTN, K_CONV_POS, K_DEC_SELF_ATTN, K_DEC_CROSS_ATTN, K_SCAN, K_ARGMAX,
 K_MISC, K_LINEAR_SKINNY, K_LINEAR_TILE64, K_DEC_QKV_CHAIN, K_DEC_PROJ_CHAIN, K_DEC_FFN_CHAIN, K_DEC_ATTN_CHAIN,
 K_DEC_VOCAB_CHAIN, K_COUNT) = range(17)
KERNEL_CLASS_NAMES = ["linear", "layernorm", "emformer_attention", "conv_pos", "decoder_self_attention",
                      "decoder_cross_attention", "scan", "argmax", "misc", "linear_skinny", "linear_tile64",
                      "dec_qkv_chain", "dec_proj_chain", "dec_ffn_chain", "dec_attn_proj_chain", "dec_vocab_chain"]

ATTN_ENUM = {"hard_aligned": ATTN_HARD, "infinite_lookback": ATTN_INFINITE_LOOKBACK,
             "waitk": ATTN_WAITK, "chunkwise": ATTN_CHUNKWISE}


class LinearDesc(C.Structure):
    _fields_ = [("M_batches", C.c_int32), ("rows_per_batch", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("a_batch_stride", C.c_int64), ("a_row_stride", C.c_int64), ("a_lead", C.c_int64),
                ("c_batch_stride", C.c_int64), ("c_row_stride", C.c_int64),
                ("r_batch_stride", C.c_int64), ("r_row_stride", C.c_int64),
                ("epilogue", C.c_int32), ("dtype", C.c_int32), ("scale", C.c_float),
                ("n_main", C.c_int32), ("aux_rows", C.c_int32), ("aux_batch_stride", C.c_int64),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("w_fragment_major", C.c_int32),
                ("c_head_dim", C.c_int32), ("c_head_stride", C.c_int64),
                ("c_tensor_heads", C.c_int32), ("c_tensor_stride", C.c_int64)]


class EmfAttnDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "T", "D", "H", "S", "R", "Lc", "M", "n_mem", "n_seg",
                                         "use_summary", "dtype")]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class DecLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "ln3_g",
                                          "ln3_b", "c_wq", "c_bq", "c_wq_soft", "c_bq_soft", "c_wo", "c_bo", "fc1",
                                          "b1", "fc2", "b2")] + [("energy_bias", C.c_float)] + \
               [(n, C.c_void_p) for n in ("k_cache", "v_cache", "head_step", "head_read", "Kmono", "Ksoft", "V", "Kpool")]


class DecoderDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "D", "H", "F", "V", "n_layers", "cap", "S_cap", "dtype", "attn_type",
                                         "ratio", "waitk_k", "mass_preservation", "online", "pad_idx", "eos_idx",
                                         "n_prev_uniform")] + \
               [("embed_scale", C.c_float)] + \
               [(n, C.c_void_p) for n in ("E", "out_proj", "pos_table", "ln_g", "ln_b", "enc_len", "n_prev", "x", "qkv",
                                          "ctx", "q", "q2", "hidden", "logits", "x_mid", "partial_self")] + \
               [("weights_fragment_major", C.c_int32), ("ffn_partial", C.c_void_p), ("P_cap", C.c_int32), ("ffn_sem", C.c_void_p)]

class CifDecLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "ln3_g", "ln3_b",
                                          "c_wq", "c_wo", "c_bo", "fc1", "b1", "fc2", "b2", "k_cache", "v_cache", "Kc")]


class CifDecoderDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "D", "H", "F", "V", "n_layers", "cap", "n_cap", "dtype", "pad_idx", "eos_idx",
                                         "highway", "n_prev_uniform")] + \
               [("embed_scale", C.c_float), ("overshoot_weight", C.c_float)] + \
               [(n, C.c_void_p) for n in ("E", "out_proj", "pos_table", "ln_g", "ln_b", "cif_len", "cif", "n_prev", "x", "qkv",
                                          "ctx", "q", "hidden", "logits", "kk", "cif_t", "eos_bias", "x_mid", "ffn_partial")] + \
               [("weights_fragment_major", C.c_int32)]


class CifStreamCtl(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("online", "done", "delays_ms", "hyp")] + \
               [(n, C.c_int32) for n in ("cap", "cur_ms", "max_len_now", "n_chunks")] + \
               [(n, C.c_void_p) for n in ("sched_cif_len", "sched_ms", "sched_max_len", "chunk_idx", "cif_len", "tok_chunk", "row_chunks")]


class StreamCtl(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("active", "read_flag", "online", "done", "delays_ms", "hyp")] + \
               [(n, C.c_int32) for n in ("cap", "cur_ms", "max_len_now", "n_chunks")] + \
               [(n, C.c_void_p) for n in ("sched_rows", "sched_ms", "sched_max_len", "chunk_idx", "enc_len", "tok_chunk", "row_chunks")] + \
               [(n, C.c_int32) for n in ("ff_waitk", "ff_ratio")] + \
               [(n, C.c_void_p) for n in ("p_probe", "step_probe", "step_force")] + [("probe_P", C.c_int32)] + \
               [("row_map", C.c_void_p), ("compact_rows", C.c_int32)]


# name -> argtypes (restype is int unless noted); mirrors include/simulst_hip.h one to one
SIGNATURES = {
    "simulst_create": [C.POINTER(_vp), _vp],
    "simulst_destroy": [_vp],
    "simulst_set_stream": [_vp, _vp],
    "simulst_last_error": [_vp],
    "simulst_version": [],
    "simulst_timer_enable": [_vp, C.c_int, C.c_int],
    "simulst_timer_read": [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(_i64)],
    "simulst_timer_reset": [_vp],
    "simulst_graph_enable": [_vp, C.c_int],
    "simulst_set_option": [_vp, _i32, _i32],
    "simulst_get_option": [_vp, _i32, C.POINTER(_i32)],
    "simulst_stream_create": [C.POINTER(_vp), _i32, _vp, _i32],
    "simulst_stream_destroy": [_vp],
    "simulst_pack_fragment_major": [_vp, _vp, _vp, _i32, _i32, _i32],
    "simulst_linear": [_vp, C.POINTER(LinearDesc), _vp, _vp, _vp, _vp, _vp, _vp],
    "simulst_conv_pos": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32],
    "simulst_ctc_best_alignment_scratch_bytes": [_i32, _i32, _i32],
    "simulst_ctc_best_alignment": [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp,
                                   _vp, _vp],
    "simulst_fbank": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_float, _i32],
    "simulst_conv_pos_mfma": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32],
    "simulst_emformer_pack_rows": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "simulst_emformer_ffn": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32],
    "simulst_emformer_ffn_prenorm": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                     _i32, _i32, _i32],
    "simulst_emformer_ffn_prenorm_qkv": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                         _i32, _i32, _i32, _i32, _i32],
    "simulst_emformer_qkv_mem_sum": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "simulst_emformer_prenorm": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "simulst_layernorm": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _i32],
    "simulst_segment_mean": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i32, _i32, _i32],
    "simulst_emformer_attention": [_vp, C.POINTER(EmfAttnDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "simulst_waitk_p_choose": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32],
    "simulst_mma_step_search": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    "simulst_expected_alignment": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32],
    "simulst_mass_preservation": [_vp, _vp, _vp, _i32, _i32, _i32],
    "simulst_expected_delays": [_vp, _vp, _vp, _i64, _i32],
    "simulst_latency_metric": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    "simulst_expected_alignment_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32],
    "simulst_expected_delays_backward": [_vp, _vp, _vp, _i64, _i32],
    "simulst_latency_metric_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    "simulst_expected_soft_attention": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32],
    "simulst_step_p_choose": [_vp, _vp, _vp, _f32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                              _vp, _i32, _i32],
    "simulst_cif_integrate": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _i32],
    "simulst_cif_alpha_head": [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _i64, _i32, _i32],
    "simulst_embed_tokens": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32],
    "simulst_decoder_self_attention": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32],
    "simulst_decoder_cross_attention": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                        _i32, _i32],
    "simulst_greedy_argmax": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32],
    "simulst_mma_decode": [_vp, C.POINTER(DecoderDesc), C.POINTER(DecLayer), _vp, _vp, _i32, _i32],
    "simulst_mma_stream_steps": [_vp, C.POINTER(DecoderDesc), C.POINTER(DecLayer), _vp, C.POINTER(StreamCtl), _i32],
    "simulst_step_p_choose_padded": [_vp, _vp, _vp, _f32, _vp, _vp] + [_i32] * 8 + [_f32, _i32],
    "simulst_pool_keys": [_vp, _vp, _vp, _vp] + [_i32] * 9,
    "simulst_policy_cross_attention": [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32,
                                       _i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "simulst_cif_decode": [_vp, C.POINTER(CifDecoderDesc), C.POINTER(CifDecLayer), _vp, _vp, _i32, _i32],
    "simulst_cif_stream_steps": [_vp, C.POINTER(CifDecoderDesc), C.POINTER(CifDecLayer), _vp, C.POINTER(CifStreamCtl), _i32],
    "simulst_cif_stream_append": [_vp] * 8 + [_i32, _i32, _i32, _i32, C.c_float, _i32, _i32],
    "simulst_decoder_proj_chain": [_vp] * 13 + [_i32, _i32, _i32],
    "simulst_decoder_ffn_chain": [_vp] * 14 + [_i32, _i32, _i32, _i32],
    "simulst_decoder_slab_sum_qkv": [_vp] * 10 + [_i32, _i32, _i32, _i32],
    "simulst_decoder_vocab_chain": [_vp] * 9 + [_i32] * 7 + [_vp, _i32, _i32],
    "simulst_has_experiments": [],
}

# entry points of a `make EXPERIMENTS=1` build (include/simulst_hip.h, SIMULST_EXPERIMENTS): measured-slower kernel families kept for A/B
EXPERIMENT_SIGNATURES = {
    "simulst_decoder_attn_proj_chain": [_vp] * 17 + [_i32] * 7,
}

# investigation hooks, present only in a `make DEBUG_HOOKS=1` build of the library (include/simulst_hip.h, SIMULST_DEBUG_HOOKS)
DEBUG_SIGNATURES = {
    "simulst_debug_ffn_variant": [_vp, C.c_int],
    "simulst_debug_chain_lds_bytes": [_vp, _i32],
    "simulst_debug_chain_xmode": [_vp, _i32],
    "simulst_debug_chain_tail": [_vp, _vp],
    "simulst_debug_chain_probe_bytes": [_i32],
    "simulst_debug_chain_probe": [_vp] * 10 + [_i32, _i32, _vp],
}
(OPT_VALU_ATTENTION, OPT_UNFUSED_DECODE, OPT_FFN_WAVES, OPT_DEC_CHAIN, OPT_DEC_ATTN_CHAIN_MAX_ROWS, OPT_DEC_ATTN_CHAIN_ROWS,
 OPT_FUSED_ARGMAX, OPT_DEC_VOCAB_CHAIN_SPLIT, OPT_DEC_EMBED_QKV_CHAIN, OPT_PANEL_WIDE, OPT_DEC_FUSE_PROJ_CROSS,
 OPT_WEIGHT_STATIONARY, OPT_CONV_TILE256, OPT_DEC_FUSE_FFN_QKV, OPT_DEC_CHAIN_ROWS32) = range(15)

_lib = None
ABI_VERSION = 108          # simulst_version(): bumped whenever a descriptor structure changes (csrc/handle.cpp)


def load():
    """Load the shared library (once). Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"simulst_amd: {LIB_PATH} is missing -- build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C simulst_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.argtypes = argtypes
        fn.restype = (C.c_char_p if name == "simulst_last_error"
                      else C.c_int64 if name == "simulst_ctc_best_alignment_scratch_bytes"
                      else C.c_int)
    for name, argtypes in EXPERIMENT_SIGNATURES.items():     # only an EXPERIMENTS build has them
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = argtypes
            fn.restype = C.c_int
    for name, argtypes in DEBUG_SIGNATURES.items():          # only a DEBUG_HOOKS build has them
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = argtypes
            fn.restype = C.c_int64 if name == "simulst_debug_chain_probe_bytes" else C.c_int
    if lib.simulst_version() != ABI_VERSION:          # the ctypes structures above mirror ONE layout of the descriptors
        raise RuntimeError(f"simulst_amd: {LIB_PATH} reports ABI version {lib.simulst_version()}, this binding is written for "
                           f"{ABI_VERSION} -- rebuild the library (make -C simulst_amd/csrc)")
    _lib = lib
    return lib


def has_experiments():
    """True when the loaded library is a `make EXPERIMENTS=1` build (tests of the measured-slower kernel families skip otherwise)"""
    return bool(load().simulst_has_experiments())


def cu_mask_words(cus_per_xcd, n_xcd=8, cus_total_per_xcd=32, take_high=False):
    """Mask words for simulst_stream_create: `cus_per_xcd` compute units of EVERY XCD (bit i = unit i // 8 of XCD i % 8 on MI355X) --
    the low-numbered units, or with take_high the high-numbered ones (the complement of the low cus_total_per_xcd - cus_per_xcd)."""
    lo, hi = (cus_total_per_xcd - cus_per_xcd, cus_total_per_xcd) if take_high else (0, cus_per_xcd)
    words = [0] * (n_xcd * cus_total_per_xcd // 32)
    for cu in range(lo, hi):
        for x in range(n_xcd):
            i = cu * n_xcd + x
            words[i >> 5] |= 1 << (i & 31)
    return words


def create_stream(device=None, cu_mask=None, priority=0):
    """A torch stream over a HIP stream made by simulst_stream_create (compute-unit mask words or a priority).  The HIP stream lives
    as long as the process (a handful per pipeline object)."""
    import torch
    lib = load()
    arr = (C.c_uint32 * len(cu_mask))(*cu_mask) if cu_mask else None
    out = _vp()
    rc = lib.simulst_stream_create(C.byref(out), int(priority), arr, len(cu_mask) if cu_mask else 0)
    if rc != 0:
        raise RuntimeError(f"simulst_stream_create failed (status {rc})")
    return torch.cuda.ExternalStream(out.value, device=device)


def destroy_stream(stream):
    """simulst_stream_destroy for a stream made by create_stream (the caller has synchronised it and drops the torch object)"""
    rc = load().simulst_stream_destroy(_vp(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError(f"simulst_stream_destroy failed (status {rc})")


class Handle:
    """Owns a simulst_handle bound to a HIP stream (default: torch's current stream)."""

    def __init__(self, stream_ptr=None):
        self.lib = load()
        if stream_ptr is None:
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError("simulst_amd: no HIP device visible; the hot path is GPU only "
                                   "(no CPU fallback)")
            stream_ptr = torch.cuda.current_stream().cuda_stream
        self._h = _vp()
        rc = self.lib.simulst_create(C.byref(self._h), _vp(stream_ptr))
        if rc != 0:
            raise RuntimeError(f"simulst_create failed: {rc}")

    @property
    def ptr(self):
        return self._h

    def set_stream(self, stream_ptr):
        self.check(self.lib.simulst_set_stream(self._h, _vp(stream_ptr)), "simulst_set_stream")

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.simulst_last_error(self._h)
            raise RuntimeError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")

    def set_option(self, option, value):
        """simulst_set_option: path selection / tuning values of this handle (OPT_* above, include/simulst_hip.h)"""
        self.check(self.lib.simulst_set_option(self._h, int(option), int(value)), "simulst_set_option")

    def get_option(self, option):
        v = _i32(0)
        self.check(self.lib.simulst_get_option(self._h, int(option), C.byref(v)), "simulst_get_option")
        return v.value

    def timer_enable(self, kernel_class=-1, on=True):
        self.check(self.lib.simulst_timer_enable(self._h, kernel_class, int(on)), "simulst_timer_enable")

    def graph_enable(self, on=True):
        self.check(self.lib.simulst_graph_enable(self._h, int(on)), "simulst_graph_enable")

    def timer_reset(self):
        self.check(self.lib.simulst_timer_reset(self._h), "simulst_timer_reset")

    def timer_read(self, kernel_class):
        ms, n = C.c_double(0), _i64(0)
        self.check(self.lib.simulst_timer_read(self._h, kernel_class, C.byref(ms), C.byref(n)), "simulst_timer_read")
        return ms.value, n.value

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.lib.simulst_destroy(self._h)
                self._h = _vp()
        except Exception:
            pass
