"""ctypes binding of libvct_hip.so (include/vct_hip.h).  The product path has NO fallback: if the
library is missing or a call fails this module raises."""
import ctypes as C
import os

import torch  # must be imported first: libvct_hip.so binds to the HIP runtime torch already loaded

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libvct_hip.so")
_AB_LIB = os.environ.get("VCT_LIB_PATH")      # developer A/B: load another build of the SAME ABI (tools/ab_build.sh)

F32, BF16 = 0, 1
ABI_VERSION = 15
GEMM_GROUP_MAX = 8
ACT = {"none": 0, None: 0, "gelu": 1, "relu": 2}
_ERR = {-1: "VCT_E_ARG (null pointer / bad enum)", -2: "VCT_E_SHAPE (unsupported shape)",
        -3: "VCT_E_ALIGN (leading dimension / alignment)", -4: "VCT_E_WORKSPACE (workspace too small)"}

vp, i32, i64, u32, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float


class GemmAdam(C.Structure):
    """include/vct_hip.h, vct_gemm_adam: optimizer epilogue of a weight-gradient GEMM."""
    _fields_ = [("param", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("shadow", vp), ("ld_shadow", i64),
                ("pk_stream", vp), ("pk_K", i32), ("pk_mode", i32), ("pk_chunk0", i32 * 4), ("pk_row0", i32), ("store_grad", i32),
                ("hyper", vp), ("step", vp)]


class AdamRange(C.Structure):
    _fields_ = [("begin", i64), ("end", i64), ("blk0", i32), ("shadow", i32)]


class GemmDesc(C.Structure):
    _fields_ = [("dtype", i32), ("out_dtype", i32), ("ta", i32), ("tb", i32), ("M", i32), ("N", i32), ("K", i32),
                ("act", i32), ("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp), ("ldc", i64),
                ("bias", vp), ("preact", vp), ("ld_preact", i64), ("addend", vp), ("ld_addend", i64),
                ("dact_src", vp), ("ld_dact", i64), ("seed", vp), ("site", u32), ("p_drop", f32),
                ("bias_grad", vp), ("workspace", vp), ("workspace_bytes", i64), ("split_k", i32), ("reserved", i32),
                ("n_tile_counters", i32), ("tile_counters", vp), ("adam", C.POINTER(GemmAdam))]


class AttnDesc(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("H", i32), ("Lq", i32), ("Lk", i32), ("hd", i32), ("causal", i32),
                ("key_pad_shift", i32), ("q", vp), ("ldq", i64), ("k", vp), ("ldk", i64), ("v", vp), ("ldv", i64),
                ("o", vp), ("ldo", i64), ("key_pad", vp), ("seed", vp), ("site", u32), ("p_drop", f32),
                ("d_o", vp), ("ld_do", i64), ("dq", vp), ("ld_dq", i64), ("dk", vp), ("ld_dk", i64),
                ("dv", vp), ("ld_dv", i64), ("q_bs", i64), ("k_bs", i64), ("v_bs", i64), ("o_bs", i64),
                ("key_ids", vp), ("key_ids_bs", i64), ("pad_id", i64)]


class SsNorm(C.Structure):
    _fields_ = [("gamma", vp), ("beta", vp), ("y", vp), ("mean", vp), ("rstd", vp)]


class LayerSsDesc(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("L", i32), ("Lm", i32), ("d", i32), ("H", i32), ("ff", i32), ("act", i32),
                ("last", i32), ("causal", i32), ("key_pad_shift", i32), ("reserved", i32),
                ("wpk", vp), ("nchunks", i64), ("x", vp), ("mem", vp), ("b_qkv", vp), ("b_o", vp), ("qkv", vp), ("o", vp), ("a", vp),
                ("n1", SsNorm), ("b_cq", vp), ("b_ckv", vp), ("b_co", vp), ("cq", vp), ("ckv", vp), ("co", vp), ("ca", vp),
                ("n2", SsNorm), ("b1", vp), ("b2", vp), ("hpre", vp), ("h", vp), ("f", vp), ("n3", SsNorm), ("nf", SsNorm),
                ("key_pad", vp), ("key_ids", vp), ("key_ids_bs", i64), ("pad_id", i64), ("seed", vp), ("p_drop", f32),
                ("site_sa", u32), ("site_n1", u32), ("site_ca", u32), ("site_n2", u32), ("site_ff", u32), ("site_n3", u32), ("site_emb", u32),
                ("pro", i32), ("feats_dtype", i32), ("feats", vp), ("x_in", vp), ("b_unify", vp), ("pe_rows", vp),
                ("emb_ids", vp), ("emb_ids_bs", i64), ("emb_table", vp), ("emb_pos", vp)]


class SsPackSeg(C.Structure):
    _fields_ = [("w", vp), ("ldw", i64), ("nchunks", i32), ("transposed", i32), ("dst_chunk", i64)]


class AdamPackSeg(C.Structure):
    _fields_ = [("begin", i64), ("end", i64), ("K", i32), ("mode", i32), ("chunk0", i32 * 4), ("stream", vp)]


class SsBwdNorm(C.Structure):
    _fields_ = [("gamma", vp), ("mean", vp), ("rstd", vp), ("ws", vp)]


class LayerSsBwdDesc(C.Structure):
    _fields_ = [("dtype", i32), ("B", i32), ("L", i32), ("d", i32), ("H", i32), ("ff", i32), ("act", i32), ("last", i32), ("causal", i32),
                ("key_pad_shift", i32), ("wpk", vp), ("nchunks", i64), ("dy", vp), ("dx", vp), ("y_last", vp),
                ("x", vp), ("qkv", vp), ("a", vp), ("x1", vp), ("hpre", vp), ("f", vp),
                ("n1", SsBwdNorm), ("n3", SsBwdNorm), ("nf", SsBwdNorm),
                ("df", vp), ("dhpre", vp), ("da", vp), ("dqkv", vp),
                ("key_pad", vp), ("key_ids", vp), ("key_ids_bs", i64), ("pad_id", i64), ("seed", vp), ("p_drop", f32),
                ("site_sa", u32), ("site_n1", u32), ("site_ff", u32), ("site_n3", u32)]


class DecodeGemvDesc(C.Structure):
    _fields_ = [("wdtype", i32), ("B", i32), ("N", i32), ("K", i32), ("W", vp), ("ldw", i64), ("bias", vp), ("pro", i32), ("act", i32),
                ("x_in", vp), ("ld_x", i64), ("g1", vp), ("b1", vp), ("g2", vp), ("b2", vp),
                ("ids", vp), ("id_stride", i64), ("table", vp), ("pos_row", vp),
                ("q", vp), ("q_bs", i64), ("kc", vp), ("vc", vp), ("kv_ld", i64), ("kv_bs", i64), ("H", i32), ("Lk", i32),
                ("res", vp), ("ld_res", i64), ("out", vp), ("ld_out", i64), ("out_native", i32), ("rows_per_wave", i32),
                ("x_out", vp)]


class DecodeLinearDesc(C.Structure):
    _fields_ = [("M", i32), ("N", i32), ("K", i32), ("out_dtype", i32), ("act", i32), ("res_dtype", i32),
                ("x", vp), ("ldx", i64), ("x_pre", vp), ("ld_pre", i64), ("ln_g", vp), ("ln_b", vp), ("x_norm", vp), ("ld_norm", i64),
                ("W", vp), ("ldw", i64), ("bias", vp), ("res", vp), ("ld_res", i64), ("out", vp), ("ldo", i64),
                ("ids", vp), ("id_stride", i64)]


class DecodeBlockDesc(C.Structure):
    _fields_ = [("kind", i32), ("d", i32), ("ff", i32), ("V", i32), ("Lk", i32), ("act", i32),
                ("id", vp), ("table", vp), ("pos_row", vp), ("res", vp), ("res_bias", vp), ("part", vp), ("n_part", i32), ("pad0", i32),
                ("g1", vp), ("b1", vp), ("g2", vp), ("b2", vp), ("x_out", vp), ("w_a", vp), ("ld_a", i64), ("b_a", vp), ("slot", vp),
                ("kc", vp), ("vc", vp), ("kv_ld", i64), ("w_b", vp), ("ld_b", i64), ("part_out", vp),
                ("sel_ws", vp), ("tok_out", vp), ("end_id", i64), ("ended", vp), ("ended_count", vp), ("all_ended_at", vp), ("t", i32), ("pad1", i32)]


DEC_PRO = {"none": 0, "embed": 1, "ln": 2, "ln_ln": 3, "self_attn": 4, "cross_attn": 5}

_SIGS = {
    "vct_abi_version": (C.c_int, []),
    "vct_build_info": (C.c_int, [C.c_char_p, C.c_int]),
    "vct_gemm": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "vct_gemm_workspace_bytes": (i64, [C.POINTER(GemmDesc)]),
    "vct_gemm_grouped": (C.c_int, [C.POINTER(GemmDesc), i32, vp]),
    "vct_gemm_grouped_workspace_bytes": (i64, [C.POINTER(GemmDesc), i32, i32]),
    "vct_attn_fwd": (C.c_int, [C.POINTER(AttnDesc), vp]),
    "vct_attn_bwd": (C.c_int, [C.POINTER(AttnDesc), vp]),
    "vct_layer_ss_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vct_layer_ss_stream_chunks": (i64, [C.c_int, C.c_int]),
    "vct_ss_pack": (C.c_int, [C.POINTER(SsPackSeg), C.c_int, vp, vp]),
    "vct_layer_ss_fwd": (C.c_int, [C.POINTER(LayerSsDesc), C.c_int, vp]),
    "vct_layer_ss_bwd_stream_chunks": (i64, [C.c_int]),
    "vct_layer_ss_bwd": (C.c_int, [C.POINTER(LayerSsBwdDesc), C.c_int, vp]),
    "vct_add_ln_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, u32, f32, vp]),
    "vct_add_ln_ln_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, f32, vp]),
    "vct_add_ln_ln_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, f32, vp]),
    "vct_add_ln_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, f32, vp]),
    "vct_ln_ws_rows": (C.c_int, [C.c_int]),
    "vct_ln_param_finalize_batched": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "vct_enc_frontend_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "vct_enc_frontend_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "vct_embed_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, vp, vp, vp, u32, f32, vp]),
    "vct_embed_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, i64, vp, vp, vp, i64, C.c_int, vp, u32, f32, vp]),
    "vct_sce_loss": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, i64, f32, vp, vp, i64, vp, vp]),
    "vct_warm": (C.c_int, [vp, i64, vp]),
    "vct_cast": (C.c_int, [C.c_int, C.c_int, vp, vp, i64, vp]),
    "vct_argmax_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, vp]),
    "vct_transpose": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, vp]),
    "vct_decode_gemv": (C.c_int, [C.POINTER(DecodeGemvDesc), vp]),
    "vct_decode_block_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vct_decode_block": (C.c_int, [C.POINTER(DecodeBlockDesc), vp]),
    "vct_decode_linear": (C.c_int, [C.POINTER(DecodeLinearDesc), vp]),
    "vct_decode_ln2": (C.c_int, [C.c_int, C.c_int, vp, i64, vp, vp, vp, vp, vp, i64, vp]),
    "vct_advance_seed": (C.c_int, [vp, vp]),
    "vct_greedy_select": (C.c_int, [C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, i64, vp, vp, vp, i32, vp]),
    "vct_gather_pad_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "vct_adam_step": (C.c_int, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, i64, i64, i32, vp, vp]),
    "vct_adam_step_pk": (C.c_int, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, i64, i64, i32, vp, vp, C.c_int, i64, vp]),
    "vct_adam_step_ranges": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, f32, f32, vp, vp, vp, i32, vp]),
    "vct_adam_step_2d": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i64, f32, f32, f32, f32, f32, vp, vp, vp]),
    "vct_cmdlist_create": (C.c_int, [C.POINTER(vp)]),
    "vct_cmdlist_destroy": (C.c_int, [vp]),
    "vct_cmdlist_begin": (C.c_int, [vp, vp]),
    "vct_cmdlist_end": (C.c_int, [vp]),
    "vct_cmdlist_replay": (C.c_int, [vp, vp]),
    "vct_cmdlist_size": (C.c_int, [vp]),
    "vct_cmdlist_streams": (C.c_int, [vp]),
    "vct_cmdlist_inject_status": (C.c_int, [C.c_int, vp]),
    "vct_cmdlist_host_call": (C.c_int, [vp, vp, vp]),
    "vct_stream_wait": (C.c_int, [vp, vp]),
    "vct_sync_record": (C.c_int, [C.c_int, vp]),
    "vct_sync_wait": (C.c_int, [C.c_int, vp]),
    "vct_comm_available": (C.c_int, []),
    "vct_comm_unique_id": (C.c_int, [vp]),
    "vct_comm_init": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "vct_comm_destroy": (C.c_int, [vp]),
    "vct_comm_rank": (C.c_int, [vp]),
    "vct_comm_world": (C.c_int, [vp]),
    "vct_comm_stream": (C.c_int, [vp, C.POINTER(vp)]),
    "vct_comm_allreduce_avg": (C.c_int, [vp, vp, i64, C.c_int, vp, C.c_int]),
    "vct_comm_reduce_scatter_avg": (C.c_int, [vp, vp, i64, C.c_int, vp, C.c_int]),
    "vct_comm_all_gather": (C.c_int, [vp, vp, i64, C.c_int, vp, C.c_int]),
    "vct_comm_broadcast": (C.c_int, [vp, vp, i64, C.c_int, C.c_int, vp, C.c_int]),
    "vct_comm_wait": (C.c_int, [vp, vp]),
    "vct_stream_create_masked": (C.c_int, [C.POINTER(u32), C.c_int, C.POINTER(vp)]),
    "vct_stream_destroy": (C.c_int, [vp]),
    "vct_tap_enable": (C.c_int, [C.c_int]),
    "vct_tap": (C.c_int, [C.c_int, C.c_int, vp]),
    "vct_tap_collect": (C.c_int, [C.c_int, C.POINTER(f32), C.c_int]),
}
_OPTIONAL = {}
HOST_FN = C.CFUNCTYPE(C.c_int, vp)        # int fn(void* arg): vct_cmdlist_host_call

_lib = None


def exported_symbols():
    """Every symbol include/vct_hip.h declares (used by the CPU-side ABI test)."""
    return sorted(list(_SIGS) + list(_OPTIONAL))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if _AB_LIB:
        # developer A/B library (tools/ab_build.sh): no stamp (it is built from an older header on purpose), but the same ABI gate as
        # the in-tree library -- a .so with another entry-point table must not be driven through this ctypes table
        lib = C.CDLL(_AB_LIB)
        for name, (res, args) in _SIGS.items():
            if not hasattr(lib, name):
                raise RuntimeError(f"VCT_LIB_PATH={_AB_LIB}: symbol {name} is missing (built from another include/vct_hip.h)")
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.vct_abi_version() != ABI_VERSION:
            raise RuntimeError(f"VCT_LIB_PATH={_AB_LIB}: ABI version {lib.vct_abi_version()} != {ABI_VERSION}")
        _lib = lib
        return lib
    _ensure_current()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is the product path and has no CPU/eager fallback. "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in _OPTIONAL.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    if lib.vct_abi_version() != ABI_VERSION:
        raise RuntimeError("libvct_hip.so ABI version mismatch")
    # one HIP runtime per process: our kernels must launch on torch's streams
    try:
        n = sum(1 for ln in open("/proc/self/maps") if "libamdhip64" in ln and " r-xp " in ln)
        if n > 1:
            raise RuntimeError("two copies of libamdhip64 are mapped; import torch before loading libvct_hip.so")
    except OSError:
        pass
    _lib = lib
    return lib


def _ensure_current():
    """The in-tree libvct_hip.so must match csrc/: if the build stamp is stale (sources edited after the last
    build) and hipcc is available, rebuild; never fall back to anything else."""
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("_vct_build", os.path.join(_PKG, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        stamp = os.path.join(_PKG, "build", "stamp")
        fresh = os.path.exists(LIB_PATH) and mod._read(stamp) == mod._digest()
        if not fresh and os.path.exists(mod._hipcc()):
            import fcntl
            import sys
            os.makedirs(os.path.join(_PKG, "build"), exist_ok=True)
            with open(os.path.join(_PKG, "build", "lock"), "w") as lk:     # one rank of a multi-process launch builds
                fcntl.flock(lk, fcntl.LOCK_EX)
                try:
                    if not (os.path.exists(LIB_PATH) and mod._read(stamp) == mod._digest()):
                        print("[vct_amd] libvct_hip.so is missing or older than csrc/: rebuilding with hipcc ...", file=sys.stderr)
                        mod.build_library()
                finally:
                    fcntl.flock(lk, fcntl.LOCK_UN)
        elif not fresh and os.path.exists(LIB_PATH):
            raise RuntimeError("csrc/ changed after libvct_hip.so was built and hipcc is not available to rebuild it")
    except Exception as e:   # never fall through to a library that no longer matches csrc/
        raise RuntimeError(f"libvct_hip.so is missing or stale and rebuilding it failed: {e}") from e


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise ValueError(f"{what}: {_ERR.get(rc, rc)}")
    if rc >= 10000:
        raise RuntimeError(f"{what}: ncclResult_t {rc - 10000}")
    raise RuntimeError(f"{what}: hipError_t {rc}")


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()
