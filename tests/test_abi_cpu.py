"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU and exports every
symbol include/vct_hip.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "vct_hip.h")


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__ as g
    g.build()
    from vct_amd import _lib
    return _lib.LIB_PATH


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vct_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib_path):
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    exported = set(l.split()[-1] for l in out.splitlines() if l.strip())
    decl = declared_symbols()
    assert len(decl) >= 17
    missing = [s for s in decl if s not in exported]
    assert not missing, missing


def test_binding_covers_header(lib_path):
    from vct_amd import _lib
    assert sorted(set(declared_symbols())) == sorted(set(_lib.exported_symbols()))
    lib = _lib.load()
    assert lib.vct_abi_version() == _lib.ABI_VERSION == 15
    buf = ctypes.create_string_buffer(128)
    assert lib.vct_build_info(buf, 128) > 0 and b"gfx950" in buf.value


def test_argument_errors_are_codes_not_crashes(lib_path):
    """Null descriptors / bad enums return VCT_E_* (negative) without touching a device."""
    from vct_amd import _lib
    lib = _lib.load()
    assert lib.vct_gemm(None, None) == -1
    d = _lib.GemmDesc()
    assert lib.vct_gemm(d, None) == -1                      # null pointers
    assert lib.vct_attn_fwd(None, None) == -1
    assert lib.vct_cast(7, 0, None, None, 10, None) == -1   # bad dtype enum
    assert lib.vct_ln_ws_rows(4864) > 0
    with pytest.raises(ValueError):
        _lib.check(-2, "x")
    with pytest.raises(RuntimeError):
        _lib.check(700, "x")


def test_new_entry_points_validate_arguments(lib_path):
    from vct_amd import _lib
    lib = _lib.load()
    assert lib.vct_gemm_grouped(None, 3, None) == -1
    arr = (_lib.GemmDesc * 9)()
    assert lib.vct_gemm_grouped(arr, 9, None) == -1                    # more than VCT_GEMM_GROUP_MAX problems
    assert lib.vct_gemm_grouped(arr, 2, None) == -1                    # null operands inside the descriptors
    assert lib.vct_gemm_grouped_workspace_bytes(arr, 2, 5) == 0
    assert lib.vct_greedy_select(1, 4, 10, None, 16, None, 1, 102, None, None, None, 3, None) == -1
    assert lib.vct_gather_pad_rows(0, 4, 3, 16, None, None, None, None, None, None) == -1
    assert lib.vct_warm(None, 64, None) == -1 and lib.vct_warm(ctypes.c_void_p(16), 8, None) == -2
    assert lib.vct_add_ln_ln_bwd(1, 8, 64, *([None] * 15), 0, 0.0, None) == -1


def test_descriptor_structs_match_the_c_header(tmp_path):
    """ctypes mirrors of the descriptor structs must have the C compiler's size and field offsets."""
    from vct_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "vct_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(vct_gemm_desc), offsetof(vct_gemm_desc, workspace), offsetof(vct_gemm_desc, tile_counters),
         sizeof(vct_attn_desc), offsetof(vct_attn_desc, d_o), offsetof(vct_attn_desc, q_bs));
  printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vct_layer_ss_desc), offsetof(vct_layer_ss_desc, wpk), offsetof(vct_layer_ss_desc, n2),
         offsetof(vct_layer_ss_desc, key_pad), offsetof(vct_layer_ss_desc, site_n3), sizeof(vct_ss_pack_seg), offsetof(vct_ss_pack_seg, dst_chunk));
  printf("%zu %zu %zu %zu %zu\\n", sizeof(vct_layer_ss_bwd_desc), offsetof(vct_layer_ss_bwd_desc, wpk), offsetof(vct_layer_ss_bwd_desc, n3),
         offsetof(vct_layer_ss_bwd_desc, key_pad), offsetof(vct_layer_ss_bwd_desc, site_n3));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vct_gemm_adam), offsetof(vct_gemm_adam, shadow), offsetof(vct_gemm_adam, pk_stream),
         offsetof(vct_gemm_adam, pk_chunk0), offsetof(vct_gemm_adam, store_grad), offsetof(vct_gemm_adam, step), offsetof(vct_gemm_desc, adam),
         sizeof(vct_adam_range));
  return 0;
}''')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    G, A = _lib.GemmDesc, _lib.AttnDesc
    S, P = _lib.LayerSsDesc, _lib.SsPackSeg
    assert got == [ctypes.sizeof(G), G.workspace.offset, G.tile_counters.offset, ctypes.sizeof(A), A.d_o.offset, A.q_bs.offset,
                   ctypes.sizeof(S), S.wpk.offset, S.n2.offset, S.key_pad.offset, S.site_n3.offset, ctypes.sizeof(P), P.dst_chunk.offset,
                   ctypes.sizeof(_lib.LayerSsBwdDesc), _lib.LayerSsBwdDesc.wpk.offset, _lib.LayerSsBwdDesc.n3.offset,
                   _lib.LayerSsBwdDesc.key_pad.offset, _lib.LayerSsBwdDesc.site_n3.offset,
                   ctypes.sizeof(_lib.GemmAdam), _lib.GemmAdam.shadow.offset, _lib.GemmAdam.pk_stream.offset, _lib.GemmAdam.pk_chunk0.offset,
                   _lib.GemmAdam.store_grad.offset, _lib.GemmAdam.step.offset, G.adam.offset, ctypes.sizeof(_lib.AdamRange)]


def test_layer_ss_entry_points_validate_arguments(lib_path):
    """Sample-stationary layer forward (csrc/vct_layer_ss.hip): shape predicate, stream length and argument errors as codes."""
    from vct_amd import _lib
    lib = _lib.load()
    assert lib.vct_layer_ss_supported(_lib.BF16, 512, 8, 2048, 19, 13) == 1
    assert lib.vct_layer_ss_supported(_lib.BF16, 512, 8, 2048, 13, 0) == 1
    assert lib.vct_layer_ss_supported(_lib.F32, 512, 8, 2048, 19, 13) == 0          # parity mode stays on the unfused kernels
    assert lib.vct_layer_ss_supported(_lib.BF16, 768, 8, 2048, 19, 13) == 0
    assert lib.vct_layer_ss_supported(_lib.BF16, 512, 8, 2048, 33, 13) == 0 and lib.vct_layer_ss_supported(_lib.BF16, 512, 8, 2048, 19, 17) == 0
    assert lib.vct_layer_ss_supported(_lib.BF16, 512, 8, 1000, 19, 13) == 0
    assert lib.vct_layer_ss_stream_chunks(2048, 0) == 96 and lib.vct_layer_ss_stream_chunks(2048, 1) == 128     # 6.3 MB / 8.4 MB of bf16
    assert lib.vct_layer_ss_fwd(None, 1, None) == -1
    d = _lib.LayerSsDesc()
    d.dtype, d.B, d.L, d.d, d.H, d.ff = _lib.BF16, 4, 13, 512, 8, 2048
    d.nchunks = 96
    assert lib.vct_layer_ss_fwd(d, 1, None) == -1                                   # null operands
    d.nchunks = 95
    assert lib.vct_layer_ss_fwd(d, 1, None) in (-1, -2)                            # (stream length does not match the layer)
    assert lib.vct_ss_pack(None, 1, None, None) == -1
    # the backward of a self-attention + feed-forward stack (csrc/vct_layer_ss_bwd.hip)
    assert lib.vct_layer_ss_bwd_stream_chunks(2048) == 96 and lib.vct_layer_ss_bwd_stream_chunks(512) == 48
    assert lib.vct_layer_ss_bwd(None, 1, None) == -1
    q = _lib.LayerSsBwdDesc()
    q.dtype, q.B, q.L, q.d, q.H, q.ff, q.nchunks = _lib.BF16, 4, 13, 512, 8, 2048, 96
    assert lib.vct_layer_ss_bwd(q, 1, None) == -1                                   # null operands
    assert lib.vct_layer_ss_bwd(q, 5, None) == -2                                   # more layers than one launch carries
    q.d = 768
    assert lib.vct_layer_ss_bwd(q, 1, None) == -2
    # Adam that also maintains stream-order packed weight copies: a table without a shadow / a misaligned base are argument errors
    assert ctypes.sizeof(_lib.AdamPackSeg) == 48
    buf = (ctypes.c_float * 16)()
    p = ctypes.addressof(buf)
    assert lib.vct_adam_step_pk(p, p, p, p, None, 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, p, 0, 0, 0, None, p, 1, 0, None) == -1
    assert lib.vct_adam_step_pk(p, p, p, p, p, 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, p, 0, 0, 0, None, p, 1, 2, None) == -1
    assert lib.vct_adam_step_pk(p, p, p, p, p, 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, p, 0, 0, 0, None, None, -1, 0, None) == -1


def test_no_cpu_fallback_when_library_is_missing(monkeypatch, lib_path):
    from vct_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvct_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/eager fallback"):
        _lib.load()


def test_runtime_entry_points_validate_arguments(lib_path):
    """Launch lists / sync points / taps (csrc/vct_runtime.hip): argument errors are codes; an empty recording works
    without a device."""
    from vct_amd import _lib
    lib = _lib.load()
    assert lib.vct_cmdlist_create(None) == -1
    h = ctypes.c_void_p()
    assert lib.vct_cmdlist_create(ctypes.byref(h)) == 0 and h.value
    assert lib.vct_cmdlist_end(h) == -1                      # not recording
    assert lib.vct_cmdlist_begin(h, None) == 0
    assert lib.vct_cmdlist_begin(h, None) == -1              # one recording per thread
    assert lib.vct_cmdlist_replay(h, None) == -1             # still recording
    assert lib.vct_cmdlist_end(h) == 0
    assert lib.vct_cmdlist_size(h) == 0 and lib.vct_cmdlist_streams(h) == 1
    assert lib.vct_cmdlist_replay(h, None) == 0              # nothing to issue
    assert lib.vct_cmdlist_destroy(h) == 0
    assert lib.vct_sync_record(64, None) == -1 and lib.vct_sync_wait(-1, None) == -1
    assert lib.vct_stream_wait(None, None) == 0              # same stream: no edge needed
    assert lib.vct_tap(24, 0, None) == -1 and lib.vct_tap(0, 2, None) == -1
    assert lib.vct_tap(0, 0, None) == 0                      # taps disabled: no-op
    assert lib.vct_tap_collect(0, None, 0) == 0


def test_replay_reports_the_first_failed_command(lib_path):
    """A command of a recorded list cannot return a status to the call that recorded it: the first non-zero status of a replay
    (failed RCCL collective, event record / wait, memset) must come back from vct_cmdlist_replay.  vct_cmdlist_inject_status is
    the fault-injection hook of that path (host commands, which need a device to drain a stream: tests/test_executor_gpu.py)."""
    from vct_amd import _lib
    lib = _lib.load()
    assert lib.vct_cmdlist_inject_status(0, None) == 0 and lib.vct_cmdlist_inject_status(10005, None) == 10005      # eager
    assert lib.vct_cmdlist_host_call(None, None, None) == -1
    h = ctypes.c_void_p()
    assert lib.vct_cmdlist_create(ctypes.byref(h)) == 0
    assert lib.vct_cmdlist_begin(h, None) == 0
    assert lib.vct_cmdlist_inject_status(0, None) == 0
    assert lib.vct_cmdlist_inject_status(10003, None) == 0        # recorded, not reported now
    assert lib.vct_cmdlist_inject_status(10007, None) == 0
    assert lib.vct_cmdlist_end(h) == 0
    assert lib.vct_cmdlist_size(h) == 3
    assert lib.vct_cmdlist_replay(h, None) == 10003               # the FIRST failure, sticky over the rest of the replay
    with pytest.raises(RuntimeError, match="ncclResult_t 3"):
        _lib.check(lib.vct_cmdlist_replay(h, None), "replay")
    assert lib.vct_cmdlist_begin(h, None) == 0 and lib.vct_cmdlist_end(h) == 0
    assert lib.vct_cmdlist_replay(h, None) == 0                   # a fresh (empty) recording: no stale status
    assert lib.vct_cmdlist_destroy(h) == 0
