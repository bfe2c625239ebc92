"""ctypes binding of libhipt_abmil.so (C ABI: include/hipt_abmil.h).

There is NO fallback: every product forward goes through this library, and anything that
needs it raises ``RuntimeError`` if the shared object is missing or a call fails.  PyTorch is
used only for device memory (tensors, ``data_ptr``) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
# HIPT_AMD_LIB selects another build of the same ABI (the diagnostic twin libhipt_abmil_dbg.so of `make DEBUG_STAMPS=1`)
LIB_PATH = os.environ.get("HIPT_AMD_LIB") or os.path.join(HERE, "libhipt_abmil.so")
CSRC = os.path.join(HERE, "csrc")

HIPT_F32, HIPT_BF16 = 0, 1
EPI_GELU, EPI_RESID, EPI_OUT_F32, EPI_RELU = 1, 2, 4, 16
ABI_VERSION = 5
PACK_QKV, PACK_PROJ, PACK_MLP, PACK_QKV_ATT = 0, 1, 2, 3

c_f32p = C.c_void_p  # device pointers travel as integers


class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
        "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "qkv_pk", "proj_pk", "mlp_pk")] + [
        ("mlp_pk_fmt", C.c_int32), ("reserved", C.c_int32), ("qkv_att_pk", C.c_void_p)]


class VitWeights(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32),
                ("hidden", C.c_int32), ("ntok", C.c_int32), ("embed_k", C.c_int32), ("ln_eps", C.c_float),
                ("attn_scale", C.c_float), ("reserved", C.c_int32),
                ("embed_w", C.c_void_p), ("embed_b", C.c_void_p), ("cls", C.c_void_p), ("pos", C.c_void_p),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p), ("blocks", C.POINTER(BlockWeights))]


class ImageLayout(C.Structure):
    _fields_ = [("grid_w", C.c_int32), ("grid_h", C.c_int32), ("patch_h", C.c_int32), ("patch_w", C.c_int32),
                ("row_stride", C.c_int64), ("chan_stride", C.c_int64), ("batch_stride", C.c_int64)]


class ClamWeights(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("s0", C.c_int32), ("s1", C.c_int32), ("s2", C.c_int32),
                ("n_classes", C.c_int32), ("n_att", C.c_int32),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("wab", C.c_void_p), ("bab", C.c_void_p),
                ("wc", C.c_void_p), ("bc", C.c_void_p), ("wcls", C.c_void_p), ("bcls", C.c_void_p),
                ("logit_bound", C.c_float), ("reserved2", C.c_int32), ("stream_pk", C.c_void_p)]


class ClamTrainWeights(C.Structure):
    _fields_ = [("s0", C.c_int32), ("s1", C.c_int32), ("s2", C.c_int32), ("n_att", C.c_int32), ("n_classes", C.c_int32),
                ("multi_branch", C.c_int32)] + [(n, C.c_void_p) for n in ("w1", "b1", "wa", "ba", "wb", "bb", "wc", "bc", "wcls", "bcls")]


class ClamTrainGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dw1", "db1", "dwa", "dba", "dwb", "dbb", "dwc", "dbc", "dwcls", "dbcls", "dbag")]


_VW, _IL, _CW = C.POINTER(VitWeights), C.POINTER(ImageLayout), C.POINTER(ClamWeights)
_TW, _TG = C.POINTER(ClamTrainWeights), C.POINTER(ClamTrainGrads)
_i, _i64, _p, _sz, _f = C.c_int, C.c_int64, C.c_void_p, C.c_size_t, C.c_float

# name -> (restype, argtypes); mirrors include/hipt_abmil.h one to one
SIGNATURES = {
    "hipt_abi_version": (_i, []),
    "hipt_last_error": (C.c_char_p, []),
    "hipt_profile_enable": (_i, [_i]),
    "hipt_profile_categories": (_i, []),
    "hipt_profile_category_name": (C.c_char_p, [_i]),
    "hipt_profile_read": (_i, [_p, _p]),
    "hipt_layernorm": (_i, [_p, _i64, _p, _p, _p, _i, _i64, _i, _i, _f, _p]),
    "hipt_linear": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "hipt_attention": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    "hipt_vit_workspace_bytes": (_sz, [_VW, _i]),
    "hipt_vit_packed_bytes": (_sz, [_VW, _i]),
    "hipt_vit_mlp_pack_format": (_i, [_VW]),
    "hipt_vit_pack_weights": (_i, [_VW, _i, _i, _p, _p]),
    "hipt_vit256_forward_workspace_bytes": (_sz, [_VW, _IL, _i, _i]),
    "hipt_vit4k_forward_workspace_bytes": (_sz, [_VW, _i]),
    "hipt_vit256_prepare_tokens": (_i, [_VW, _p, _IL, _i, _i, _p, _p, _sz, _p]),
    "hipt_vit4k_prepare_tokens": (_i, [_VW, _p, _i, _p, _p, _sz, _p]),
    "hipt_vit_blocks": (_i, [_VW, _p, _i, _i, _i, _p, _p, _sz, _p]),
    "hipt_vit_head": (_i, [_VW, _p, _i, _i, _p, _p]),
    "hipt_vit_cls_attention": (_i, [_VW, _p, _i, _p, _p, _sz, _p]),
    "hipt_vit_attention_unit": (_i, [_VW, _i, _p, _i, _p, _i, _p, _sz, _p]),
    "hipt_vit_mlp_unit": (_i, [_VW, _i, _p, _p, _i, _p, _p, _sz, _p]),
    "hipt_vit256_forward": (_i, [_VW, _p, _IL, _i, _i, _p, _p, _sz, _p]),
    "hipt_vit4k_forward": (_i, [_VW, _p, _i, _p, _p, _sz, _p]),
    "hipt_image_compute_bytes": (_sz, [_VW, _IL, _i, _i]),
    "hipt_image_to_compute": (_i, [_VW, _p, _i, _IL, _i, _p, _p]),
    "hipt_vit256_range_workspace_bytes": (_sz, [_VW, _i, _i]),
    "hipt_vit256_forward_range": (_i, [_VW, _p, _IL, _i, _i, _i, _p, _p, _sz, _p]),
    "hipt_vit256_range_px_workspace_bytes": (_sz, [_VW, _IL, _i, _i]),
    "hipt_vit256_forward_range_px": (_i, [_VW, _p, _IL, _i, _i, _i, _p, _p, _sz, _p]),
    "hipt_hipt4k_workspace_bytes": (_sz, [_VW, _VW, _i, _i, _i, _i]),
    "hipt_hipt4k_forward": (_i, [_VW, _VW, _p, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "hipt_hipt4k_u8_workspace_bytes": (_sz, [_VW, _VW, _i, _i, _i, _i]),
    "hipt_hipt4k_forward_u8": (_i, [_VW, _VW, _p, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "hipt_u8_normalize": (_i, [_p, _i, C.c_int64, C.c_int64, _p, _i, _p]),
    "hipt_clam_workspace_bytes": (_sz, [_CW, _i]),
    "hipt_clam_stream_packed_bytes": (_sz, [_CW]),
    "hipt_clam_stream_pack": (_i, [_CW, _p, _p]),
    "hipt_clam_ticket_offset": (_sz, [_CW, _i]),
    "hipt_clam_sb_forward": (_i, [_CW, _p, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "hipt_clam_mb_supported": (_i, [_CW]),
    "hipt_clam_mb_workspace_bytes": (_sz, [_CW, _i]),
    "hipt_clam_mb_forward": (_i, [_CW, _p, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "hipt_attn_net_gated": (_i, [_CW, _p, _i, _p, _p, _sz, _p]),
    "hipt_clam_gather_h1": (_i, [_CW, _p, _p, _i, _p, _p]),
    "hipt_clam_train_workspace_bytes": (_sz, [_TW, _i]),
    "hipt_clam_train_shape_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "hipt_clam_train_forward": (_i, [_TW, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "hipt_clam_train_backward": (_i, [_TW, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _TG, _p, _sz, _p]),
    "hipt_topk_rows": (_i, [_p, _i, _i, _i, _p, _p]),
}

_lib = None
_lock = threading.Lock()
calls = 0  # number of native entry-point invocations (tests use it to prove the HIP path ran)


class NativeLibraryError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libhipt_abmil.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))],
                       capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode or not os.path.isfile(LIB_PATH):
        raise NativeLibraryError(f"building {LIB_PATH} failed (make exit {r.returncode})")
    return LIB_PATH


def lib():
    """The loaded library; raises if it is not there (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.isfile(LIB_PATH):
                raise NativeLibraryError(
                    f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                    f"(or `make -C {CSRC}`).  This package has no non-HIP execution path.")
            # PyTorch-ROCm ships its own libamdhip64: it must be in the process BEFORE this library is loaded, so that
            # the library's kernels register with the HIP runtime that owns torch's streams and memory.  Loaded the
            # other way round, the system runtime comes in first and every launch fails (hipFuncSetAttribute error).
            import torch  # noqa: F401
            h = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(h, name)  # AttributeError if the .so does not export a declared symbol
                fn.restype, fn.argtypes = res, args
            if h.hipt_abi_version() != ABI_VERSION:
                raise NativeLibraryError(f"ABI mismatch: library {h.hipt_abi_version()} != binding {ABI_VERSION}")
            _lib = h
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().hipt_last_error().decode(errors="replace")
        raise RuntimeError(f"libhipt_abmil: {what} failed with code {rc}: {msg}")


class StreamArg(C.c_void_p):
    """A hipStream_t argument that remembers which device it belongs to (see ``call``)."""
    device = None


def call(name: str, *args):
    """Invoke an int-returning entry point and raise on a non-zero status.

    The library launches on the calling thread's CURRENT device (kernel launches, hipMemsetAsync and the per-device
    hipFuncSetAttribute opt-ins all follow it), so the call is made with the device of its stream argument current:
    a module on cuda:1 while cuda:0 is current, or HIPT_4K's device256 != device4k placement (hipt_4k.py:39-46), would
    otherwise enqueue on the wrong GPU's stream against foreign memory."""
    global calls
    calls += 1
    dev = next((a.device for a in args if isinstance(a, StreamArg)), None)
    if dev is not None:
        import torch
        if dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):
                check(getattr(lib(), name)(*args), name)
            return
    check(getattr(lib(), name)(*args), name)


def stream_ptr(device=None) -> StreamArg:
    """The current HIP stream of ``device`` (default: the current device) as a call argument."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"stream_ptr: {dev} is not a HIP device")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    s = StreamArg(torch.cuda.current_stream(dev).cuda_stream)
    s.device = dev
    return s


def same_device(what: str, ref_device, *tensors) -> None:
    """Every tensor whose address is handed to a kernel must live on the device the kernel runs on: a parameter left on
    the CPU (relocate() never called) or on another GPU would be a wild pointer on the GPU -- the reference raises a
    device-mismatch RuntimeError there, and so does this."""
    for t in tensors:
        if t is not None and t.device != ref_device:
            raise RuntimeError(f"{what}: expected all tensors on {ref_device}, found one on {t.device} "
                               f"(move the module and its input to the same HIP device)")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def dtype_code(name: str) -> int:
    if name in ("fp32", "float32", "f32"):
        return HIPT_F32
    if name in ("bf16", "bfloat16"):
        return HIPT_BF16
    raise ValueError(f"compute dtype must be 'fp32' or 'bf16', got {name!r}")


def require_cuda(t, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: input is on {t.device}; hipt_abmil_atec23_amd runs only on a HIP device "
            f"(there is deliberately no CPU path — move the module and its inputs to 'cuda').")


def profile_enable(on: bool = True) -> None:
    check(lib().hipt_profile_enable(1 if on else 0), "hipt_profile_enable")


def profile_read() -> dict:
    """{category: (total_ms, launches)} since the last read (synchronises on the recorded events)."""
    n = lib().hipt_profile_categories()
    ms, cnt = (C.c_float * n)(), (C.c_int * n)()
    check(lib().hipt_profile_read(C.cast(ms, C.c_void_p), C.cast(cnt, C.c_void_p)), "hipt_profile_read")
    return {lib().hipt_profile_category_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n) if cnt[i]}
