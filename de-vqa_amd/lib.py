"""ctypes binding of csrc/libdevqa_hip.so (C ABI: include/devqa.h).

There is NO CPU fallback: if the library is missing or a call fails, this module raises.
PyTorch is used only for device memory and streams (tensor.data_ptr(), current stream).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdevqa_hip.so")

ACT_NONE, ACT_RELU, ACT_GELU, ACT_QUICK_GELU = 0, 1, 2, 3


class DevqaError(RuntimeError):
    pass


_lib = None


def _sig(fn, argtypes, restype=c_int):
    fn.argtypes = argtypes
    fn.restype = restype


def load():
    """Load the shared library (idempotent). Raises DevqaError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DevqaError("libdevqa_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback for the product path)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    P, I, F, I64 = c_void_p, c_int, c_float, c_int64
    _sig(L.devqa_last_error, [], ctypes.c_char_p)
    _sig(L.devqa_abi_version, [])
    _sig(L.devqa_gemm_bf16, [P, I64, P, I64, P, I, I, I, F, I, P, P, P, I64, P])
    _sig(L.devqa_gemm_bf16_swiglu_supported, [I, I, I])
    _sig(L.devqa_gemm_f32, [P, I64, P, I64, P, I, I, I, F, I, P, P, I64, P])
    _sig(L.devqa_rmsnorm, [P, P, P, I, I, F, P, P, P])
    _sig(L.devqa_rmsnorm_bwd_dx, [P, P, P, P, I, I, F, P, P])
    _sig(L.devqa_rope_bf16, [P, I64, I, P, I, I, F, P])
    _sig(L.devqa_rope_f32, [P, I64, I, P, I, I, F, P])
    _sig(L.devqa_swiglu_bf16, [P, I, I, P, P])
    _sig(L.devqa_swiglu_f32, [P, I, I, P, P])
    _sig(L.devqa_gemm_set_mode, [I])
    _sig(L.devqa_gemm_bf16_splitk, [P, I64, P, I64, I, I, I, I, P, P, P])
    _sig(L.devqa_profile_gemm, [I])
    _sig(L.devqa_profile_gemm_read, [P, P, P])
    _sig(L.devqa_profile, [I])
    _sig(L.devqa_profile_read, [I, P, P, P])
    _sig(L.devqa_profile_dropped, [P])
    _sig(L.devqa_layernorm, [P, P, P, P, I, I, F, P, P, P])
    _sig(L.devqa_attention_f32, [P, I64, P, I64, P, I64, P, I64, P, I, I, I, I, F, I, P])
    _sig(L.devqa_im2col_patches_f32, [P, I, I, I, I, P, P])
    _sig(L.devqa_embed_rows_f32, [P, P, P, P, P, P, I, I, I, I, I, P, P])
    _sig(L.devqa_vocab_rows_f32, [P, I64, I, I, P, P, P, P, P, I64, P])
    _sig(L.devqa_attention, [P, I64, P, I64, P, I64, P, I64, P, I, I, I, I, F, I, P])
    _sig(L.devqa_attention_reload_env, [])
    _sig(L.devqa_act_cast, [P, I, P, P, I64, P])
    _sig(L.devqa_im2col_patches, [P, I, I, I, I, P, P])
    _sig(L.devqa_vit_assemble, [P, P, P, I, I, I, P, P])
    _sig(L.devqa_embed_rows, [P, P, P, P, P, P, I, I, I, I, I, P, P])
    _sig(L.devqa_gather_rows, [P, P, I, I, I, P, P])
    _sig(L.devqa_cast_f32_bf16, [P, P, I64, P])
    _sig(L.devqa_vocab_rows, [P, I64, I, I, P, P, P, P, P, I64, P])
    _sig(L.devqa_layernorm_bwd_dx, [P, P, P, P, I, I, F, P, P])
    _sig(L.devqa_layernorm_bwd_params, [P, P, P, I, I, F, I, I, P, P, P, P])
    _sig(L.devqa_colsum_f32, [P, I, I, I, P, P])
    _sig(L.devqa_ft_adamw_step, [P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, F, F, F, F, I64, P])
    _sig(L.devqa_ft_adamw_step_fm, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, F, F, F, F, I64, P])
    _sig(L.devqa_active_columns, [P, I, I, I, P, P, P])
    _sig(L.devqa_gather_cols_f32, [P, I64, I64, I, P, I64, P, I, I, P, P])
    _sig(L.devqa_gather_cols_bf16, [P, I64, I64, I, P, I64, P, I, I, P, P])
    _sig(L.devqa_scatter_cols_add_f32, [P, I, P, P, I, P, I64, P])
    _sig(L.devqa_rows_matvec_f32, [P, I64, P, P, P, P, I, I, I, I, P])
    _sig(L.devqa_delta_op, [I, P, P, P, I64, P])
    _sig(L.devqa_ft_step_control, [P, P, I, I, I, I, F, P, P, P, P, P, P])
    _sig(L.devqa_cosine_topk_workspace, [I, I, I], c_int64)
    _sig(L.devqa_cosine_topk, [P, P, I, I, I, I, I, I, P, P, P, P])
    _sig(L.devqa_cosine_topk_cached, [P, P, P, I, I, I, I, I, P, P, P, P])
    _sig(L.devqa_row_inv_norm, [P, I, I, P, P])
    for fn in (L.devqa_attention_bwd, L.devqa_attention_bwd_f32):
        _sig(fn, [P, I64, P, I64, P, I64, P, I64, P, I64, P, I64, P, I64, P, I64, P, P, I, I, I, I, F, I, P])
    _sig(L.devqa_relu_bwd, [P, P, P, I64, P])
    _sig(L.devqa_relu_bwd_f32, [P, P, P, I64, P])
    _sig(L.devqa_gelu_f32, [P, P, P, I64, P])
    _sig(L.devqa_gelu_bwd_f32, [P, P, P, I64, P])
    _sig(L.devqa_mend_normalize_concat, [P, P, P, P, P, P, P, F, I, I, I, P, P])
    _sig(L.devqa_mend_lrlinear_epilogue, [P, P, P, P, P, P, I, I, P])
    _sig(L.devqa_logit_kl_rows, [P, I64, P, I64, I, I, P, P])
    _sig(L.devqa_kl_dlogits, [P, I64, P, I64, I, I, P, P, P, I64, I, P])
    _sig(L.devqa_welford_rows, [P, P, I, I, I, P, P, P, P, P, P])
    _sig(L.devqa_mend_lrlinear_bwd, [P, P, P, P, P, I, I, P, P, P, P, P])
    _sig(L.devqa_sumsq_f32, [P, I64, P, P])
    _sig(L.devqa_adam_step, [P, P, P, P, I64, F, F, F, F, I, P, P])
    _sig(L.devqa_swiglu_bwd_bf16, [P, P, I, I, P, P])
    _sig(L.devqa_swiglu_bwd_f32, [P, P, I, I, P, P])
    _sig(L.devqa_tp_neuron_fwd, [P, I, I, P, P, P, I, P, P, I, P, P, P])
    _sig(L.devqa_tp_neuron_bwd, [P, P, I, I, P, I, P, I, P, I, P, P, P, F, F, F, P, P, P, P, P, P])
    _sig(L.devqa_tp_gated_neuron_fwd, [P, I, I, P, P, P, I, P, P, I, P, P, P])
    _sig(L.devqa_tp_gated_neuron_bwd, [P, P, I, I, P, I, P, I, P, I, P, P, P, F, F, F, P, P, P, P, P, P])
    # ---- path level (include/devqa.h "PATH LEVEL", csrc/path_ctx.hip) ----
    U64 = ctypes.c_uint64
    _sig(L.devqa_ctx_create, [I, P, P, I, P])
    _sig(L.devqa_ctx_destroy, [U64])
    _sig(L.devqa_ctx_set_weight, [U64, ctypes.c_char_p, P])
    _sig(L.devqa_vision_encode_workspace, [U64, I], c_int64)
    _sig(L.devqa_vision_encode, [U64, P, I, P, P, I64, P])
    _sig(L.devqa_llm_layers_workspace, [U64, I, I], c_int64)
    _sig(L.devqa_llm_layers, [U64, P, P, I, I, I, I, I, I, P, P, I64, P])
    _sig(L.devqa_llm_layers_ex, [U64, P, P, P, I, I, I, I, I, I, I, P, P, I64, P])
    _sig(L.devqa_llm_prefix_workspace, [U64, I], c_int64)
    _sig(L.devqa_llm_prefix, [U64, P, P, I, I, I, I, P, P, I64, P])
    _sig(L.devqa_mend_transform_workspace, [I, I, I, I], c_int64)
    _sig(L.devqa_mend_transform, [P, P, P, I, I, I, P, P, P, P, I64, P])
    _sig(L.devqa_mend_apply_workspace, [I, I, I], c_int64)
    _sig(L.devqa_mend_apply, [P, P, P, P, I, I, I, I, I, P, I64, P])
    _sig(L.devqa_llm_head_workspace, [U64, I], c_int64)
    _sig(L.devqa_llm_head, [U64, P, P, I, P, P, I64, P])
    _sig(L.devqa_llm_forward_workspace, [U64, I, I], c_int64)
    _sig(L.devqa_llm_forward, [U64, P, P, I, I, I, I, P, I, P, P, I64, P])
    _sig(L.devqa_llm_forward_ex, [U64, P, P, P, I, I, I, I, P, I, P, P, I64, P])
    _sig(L.devqa_ft_edit_workspace, [U64, I, I, I], c_int64)
    _sig(L.devqa_ft_edit, [U64, P, I64, P, P, P, P, I, I, I, P, P, P, P, P, P, I64, P])
    _sig(L.devqa_ctx_bind_edit_target, [U64, ctypes.c_char_p, P])
    _sig(L.devqa_apply_delta, [U64, P, P])
    _sig(L.devqa_restore, [U64, P])
    _sig(L.devqa_token_acc, [P, I64, I, I, P, P, P, P, P])
    _sig(L.devqa_comm_unique_id, [P])
    _sig(L.devqa_comm_create, [I, I, P, I, P])
    _sig(L.devqa_comm_destroy, [U64])
    _sig(L.devqa_gather_scores, [U64, P, I, P, P])
    _lib = L
    return L


EXPORTS = ["devqa_ctx_create", "devqa_ctx_destroy", "devqa_ctx_set_weight", "devqa_vision_encode_workspace", "devqa_vision_encode",
           "devqa_llm_layers_workspace", "devqa_llm_layers", "devqa_llm_layers_ex", "devqa_llm_forward_ex", "devqa_llm_prefix_workspace", "devqa_llm_prefix", "devqa_mend_transform_workspace",
           "devqa_mend_transform", "devqa_mend_apply_workspace", "devqa_mend_apply", "devqa_llm_head_workspace", "devqa_llm_head", "devqa_llm_forward_workspace",
           "devqa_llm_forward", "devqa_ft_edit_workspace", "devqa_ft_edit", "devqa_ctx_bind_edit_target", "devqa_apply_delta", "devqa_restore",
           "devqa_token_acc", "devqa_comm_unique_id", "devqa_comm_create", "devqa_comm_destroy", "devqa_gather_scores",
           "devqa_profile", "devqa_profile_read", "devqa_profile_dropped", "devqa_attention_reload_env", "devqa_act_cast", "devqa_row_inv_norm", "devqa_cosine_topk_cached", "devqa_rmsnorm", "devqa_rmsnorm_bwd_dx", "devqa_rope_bf16", "devqa_rope_f32", "devqa_swiglu_bf16", "devqa_swiglu_f32",
           "devqa_gemm_bf16_splitk", "devqa_gemm_set_mode", "devqa_active_columns", "devqa_gather_cols_f32", "devqa_gather_cols_bf16", "devqa_scatter_cols_add_f32",
           "devqa_profile_gemm", "devqa_profile_gemm_read", "devqa_last_error", "devqa_abi_version", "devqa_gemm_bf16", "devqa_gemm_f32", "devqa_attention_f32",
           "devqa_im2col_patches_f32", "devqa_embed_rows_f32", "devqa_vocab_rows_f32", "devqa_layernorm", "devqa_attention",
           "devqa_im2col_patches", "devqa_vit_assemble", "devqa_embed_rows", "devqa_gather_rows", "devqa_cast_f32_bf16", "devqa_split_f32_bf16x2", "devqa_gemm_bf16_swiglu_supported",
           "devqa_vocab_rows", "devqa_layernorm_bwd_dx", "devqa_layernorm_bwd_params", "devqa_colsum_f32", "devqa_ft_adamw_step", "devqa_ft_adamw_step_fm", "devqa_rows_matvec_f32", "devqa_delta_op",
           "devqa_ft_step_control", "devqa_cosine_topk_workspace", "devqa_cosine_topk", "devqa_attention_bwd",
           "devqa_attention_bwd_f32", "devqa_relu_bwd", "devqa_relu_bwd_f32", "devqa_gelu_f32", "devqa_gelu_bwd_f32", "devqa_mend_normalize_concat",
           "devqa_mend_lrlinear_epilogue", "devqa_logit_kl_rows", "devqa_kl_dlogits", "devqa_welford_rows",
           "devqa_mend_lrlinear_bwd", "devqa_sumsq_f32", "devqa_adam_step", "devqa_tp_neuron_fwd", "devqa_tp_neuron_bwd",
           "devqa_tp_gated_neuron_fwd", "devqa_tp_gated_neuron_bwd",
           "devqa_swiglu_bwd_bf16", "devqa_swiglu_bwd_f32"]


def gemm_rows_longk(a, w):
    """fp32 [M,N] = a[M,K] @ w[N,K].T for a few rows and a very long K (dH = dlogits . E): split-K on the
    bf16 MFMA kernel, all (<= 256) rows in one launch so that w streams once (fp32 operands: exact-fp32 GEMM)."""
    if a.dtype != torch.bfloat16:
        return gemm(a, w, want="f32")
    M, K = a.shape
    N = w.shape[0]
    if M > 256:
        return gemm(a, w, want="f32")
    nk = (K + 63) // 64
    splits = max(1, min(nk, 640 // max(1, (N + 127) // 128)))
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    # rows per launch: the kernel has 64 / 128 / 256-row tiles and w streams once per launch.  Measured on dH (K = 50272, N = 2560,
    # tools/splitk_rows_bench.py): 64 rows 68 us, 100 rows 110 us (2 x 64: 138), 250 rows 234 us (4 x 64: 291); 129..192 rows are
    # cheapest as 128 + rest (the 256-row tile is compute-limited when a third of it is padding).
    env = os.environ.get("DEVQA_SPLITK_ROWS")
    if env:
        groups = [min(int(env), M - r0) for r0 in range(0, M, int(env))]
    elif 128 < M <= 192:
        groups = [128, M - 128]
    else:
        groups = [M]
    ws = torch.empty((splits, max(groups), N), dtype=torch.float32, device=a.device)
    r0 = 0
    for m in groups:
        _chk(load().devqa_gemm_bf16_splitk(_p(a[r0:r0 + m]), a.stride(0), _p(w), w.stride(0), m, N, K, splits, _p(ws), _p(out[r0:r0 + m]),
                                           _stream()), "devqa_gemm_bf16_splitk")
        r0 += m
    return out


def gemm_set_mode(mode):
    _chk(load().devqa_gemm_set_mode(int(mode)), "devqa_gemm_set_mode")


def profile_gemm(enable):
    _chk(load().devqa_profile_gemm(int(enable)), "devqa_profile_gemm")


def profile_gemm_read():
    """-> list of (ms, flops, launches) per tile variant [32x128, 64x128, 128x128, -]"""
    ms = (ctypes.c_double * 4)()
    fl = (ctypes.c_double * 4)()
    ln = (ctypes.c_int64 * 4)()
    _chk(load().devqa_profile_gemm_read(ms, fl, ln), "devqa_profile_gemm_read")
    return [(ms[i], fl[i], ln[i]) for i in range(4)]


PROF_GEMM0, PROF_ATTENTION, PROF_FT_ADAMW, PROF_COSINE, PROF_LAYERNORM = 0, 4, 5, 6, 7


def profile(enable):
    _chk(load().devqa_profile(int(enable)), "devqa_profile")


def profile_read(slot):
    """-> (ms, work, launches) of one instrumented kernel family since profile(1) (include/devqa.h)."""
    ms, wk, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _chk(load().devqa_profile_read(int(slot), ctypes.byref(ms), ctypes.byref(wk), ctypes.byref(ln)), "devqa_profile_read")
    return ms.value, wk.value, ln.value


def profile_dropped():
    """launches the slot profiler could not record since profile(1) (event pool full); 0 in a trustworthy measurement"""
    n = ctypes.c_int64()
    _chk(load().devqa_profile_dropped(ctypes.byref(n)), "devqa_profile_dropped")
    return n.value


def _chk(rc, name):
    if rc != 0:
        raise DevqaError("%s failed (%d): %s" % (name, rc, load().devqa_last_error().decode()))


def _p(t):
    return None if t is None else c_void_p(t.data_ptr())


def h2d(data, dtype, device):
    """Small host -> device transfer that does NOT wait for the stream: pinned staging + non_blocking copy.  `torch.tensor(list,
    device=...)` / pageable `.to(device)` block the host until the queued GPU work has drained (~0.2 ms per call under load), which
    serialises the per-probe bookkeeping of the generic evaluator with the GPU; the caching host allocator keeps the pinned block
    alive until the copy has run."""
    t = data if isinstance(data, torch.Tensor) else torch.as_tensor(data, dtype=dtype)
    if t.dtype != dtype:
        t = t.to(dtype)
    if not torch.device(device).type == "cuda":
        return t
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def _stream():
    # raw handle of torch's CURRENT stream on the current device, without building a torch.cuda.Stream object
    # (torch.cuda.current_stream() costs ~10 us per call; this path is called once per kernel launch)
    return c_void_p(_raw_stream(_cur_dev()))


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _cur_dev = torch._C._cuda_getDevice
except AttributeError:  # pragma: no cover - other torch builds
    def _raw_stream(_d):
        return torch.cuda.current_stream().cuda_stream

    def _cur_dev():
        return torch.cuda.current_device()


def _need(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise DevqaError("%s: expected contiguous cuda %s, got %s %s contiguous=%s" %
                         (name, dtype, t.device, t.dtype, t.is_contiguous()))


# ---------------------------------------------------------------------------------------------
# thin typed wrappers (shape bookkeeping only; every FLOP happens in the library)
# ---------------------------------------------------------------------------------------------
def gemm(a, w, bias=None, alpha=1.0, act=ACT_NONE, residual=None, out_bf16=None, out_f32=None, want="bf16"):
    """C = epi(a @ w.T). a [M,K] (row stride may exceed K), w [N,K]; both bf16 (MFMA bf16) or both
    fp32 (exact-fp32 MFMA, "faithful" mode: fp32 output only)."""
    if a.dtype == torch.float32:
        return _gemm_f32(a, w, bias, alpha, act, residual, out_f32)
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.is_cuda and w.is_cuda
    assert a.stride(-1) == 1 and w.stride(-1) == 1 and a.dim() == 2 and w.dim() == 2
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K, (a.shape, w.shape)
    if out_bf16 is None and out_f32 is None:
        if want == "bf16":
            out_bf16 = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
        elif want == "f32":
            out_f32 = torch.empty((M, N), dtype=torch.float32, device=a.device)
        else:
            out_bf16 = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
            out_f32 = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ldc = N
    for o in (out_bf16, out_f32, residual):
        if o is not None:
            assert o.shape == (M, N) and o.is_contiguous(), "gemm outputs/residual must be contiguous [M,N]"
    if bias is not None:
        _need(bias, torch.float32, "gemm bias")
        assert bias.numel() == N
    if residual is not None:
        _need(residual, torch.float32, "gemm residual")
    _chk(load().devqa_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), M, N, K, float(alpha), int(act),
                                _p(residual), _p(out_bf16), _p(out_f32), ldc, _stream()), "devqa_gemm_bf16")
    if out_bf16 is not None and out_f32 is not None:
        return out_bf16, out_f32
    return out_bf16 if out_bf16 is not None else out_f32


ACT_SWIGLU_IL16 = 4
DESC_FUSE_SWIGLU = 1


def interleave_gate_up(w_gu):
    """[2F, d] fused (gate rows | up rows) -> the row order devqa_gemm_bf16's fused SwiGLU expects: blocks of 16 gate rows followed by the 16
    matching up rows (include/devqa.h, DEVQA_ACT_SWIGLU_IL16).  F % 16 == 0."""
    F2, d = w_gu.shape
    F = F2 // 2
    assert F2 == 2 * F and F % 16 == 0
    return torch.cat([w_gu[:F].view(F // 16, 16, d), w_gu[F:].view(F // 16, 16, d)], 1).reshape(F2, d).contiguous()


def gemm_swiglu_supported(M, N, K):
    """True when a fused-SwiGLU call a [M, K] x w_il [N = 2F, K] takes the 256 x 256 kernel (devqa_gemm_bf16_swiglu_supported)."""
    return bool(load().devqa_gemm_bf16_swiglu_supported(int(M), int(N), int(K)))


def gemm_swiglu(a, w_il, out=None):
    """out [M, F] bf16 = silu(a . gate^T) * (a . up^T) with w_il = interleave_gate_up([gate | up]) -- one GEMM, no [M, 2F] intermediate."""
    assert a.dtype == torch.bfloat16 and a.is_cuda
    _need(w_il, torch.bfloat16, "gemm_swiglu w")
    M, K = a.shape
    N = w_il.shape[0]
    assert w_il.shape[1] == K and a.stride(1) == 1
    if out is None:
        out = torch.empty((M, N // 2), dtype=torch.bfloat16, device=a.device)
    assert out.shape == (M, N // 2) and out.dtype == torch.bfloat16 and out.stride(1) == 1
    _chk(load().devqa_gemm_bf16(_p(a), a.stride(0), _p(w_il), w_il.stride(0), None, M, N, K, 1.0, ACT_SWIGLU_IL16, None, _p(out), None,
                                out.stride(0), _stream()), "devqa_gemm_bf16(swiglu)")
    return out


def _gemm_f32(a, w, bias, alpha, act, residual, out_f32):
    assert w.dtype == torch.float32 and a.is_cuda and w.is_cuda and a.dim() == 2 and w.dim() == 2
    assert a.stride(-1) == 1 and w.stride(-1) == 1
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K, (a.shape, w.shape)
    if out_f32 is None:
        out_f32 = torch.empty((M, N), dtype=torch.float32, device=a.device)
    for o in (out_f32, residual):
        if o is not None:
            assert o.shape == (M, N) and o.is_contiguous() and o.dtype == torch.float32
    _chk(load().devqa_gemm_f32(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), M, N, K, float(alpha), int(act),
                               _p(residual), _p(out_f32), N, _stream()), "devqa_gemm_f32")
    return out_f32


def layernorm(x, gamma, beta, eps, add=None, want="bf16"):
    _need(x, torch.float32, "layernorm x")
    M, D = x.shape
    ob = torch.empty((M, D), dtype=torch.bfloat16, device=x.device) if want in ("bf16", "both") else None
    of = torch.empty((M, D), dtype=torch.float32, device=x.device) if want in ("f32", "both") else None
    if add is not None:
        _need(add, torch.float32, "layernorm add")
        assert add.shape == x.shape
    _chk(load().devqa_layernorm(_p(x), _p(add), _p(gamma), _p(beta), M, D, float(eps), _p(ob), _p(of), _stream()),
         "devqa_layernorm")
    if want == "both":
        return ob, of
    return ob if ob is not None else of


def rmsnorm(x, w, eps, add=None, want="bf16"):
    _need(x, torch.float32, "rmsnorm x")
    M, D = x.shape
    ob = torch.empty((M, D), dtype=torch.bfloat16, device=x.device) if want in ("bf16", "both") else None
    of = torch.empty((M, D), dtype=torch.float32, device=x.device) if want in ("f32", "both") else None
    if add is not None:
        _need(add, torch.float32, "rmsnorm add")
        assert add.shape == x.shape
    _chk(load().devqa_rmsnorm(_p(x), _p(add), _p(w), M, D, float(eps), _p(ob), _p(of), _stream()), "devqa_rmsnorm")
    if want == "both":
        return ob, of
    return ob if ob is not None else of


def rmsnorm_bwd_dx(x, w, dy, eps, add=None):
    _need(x, torch.float32, "rmsnorm_bwd x")
    _need(dy, torch.float32, "rmsnorm_bwd dy")
    M, D = x.shape
    dx = torch.empty_like(x)
    if add is not None:
        _need(add, torch.float32, "rmsnorm_bwd add")
    _chk(load().devqa_rmsnorm_bwd_dx(_p(x), _p(add), _p(w), _p(dy), M, D, float(eps), _p(dx), _stream()),
         "devqa_rmsnorm_bwd_dx")
    return dx


def rope_(x, pos, n_heads, dh, theta):
    """In-place rotary embedding on the first n_heads*dh columns of the 2-D (possibly strided) view x."""
    assert x.dim() == 2 and x.stride(1) == 1 and pos.dtype == torch.int32 and pos.numel() == x.shape[0]
    fn = load().devqa_rope_bf16 if x.dtype == torch.bfloat16 else load().devqa_rope_f32
    _chk(fn(_p(x), x.stride(0), x.shape[0], _p(pos), int(n_heads), int(dh), float(theta), _stream()), "devqa_rope")
    return x


def swiglu(gu):
    """gu [R, 2F] contiguous (gate | up) -> silu(gate) * up [R, F]"""
    assert gu.dim() == 2 and gu.is_contiguous() and gu.shape[1] % 2 == 0
    R, F2 = gu.shape
    out = torch.empty((R, F2 // 2), dtype=gu.dtype, device=gu.device)
    fn = load().devqa_swiglu_bf16 if gu.dtype == torch.bfloat16 else load().devqa_swiglu_f32
    _chk(fn(_p(gu), R, F2 // 2, _p(out), _stream()), "devqa_swiglu")
    return out


def layernorm_bwd_dx(x, gamma, dy, eps, add=None):
    _need(x, torch.float32, "ln_bwd x")
    _need(dy, torch.float32, "ln_bwd dy")
    M, D = x.shape
    dx = torch.empty_like(x)
    if add is not None:
        _need(add, torch.float32, "ln_bwd add")
        assert add.shape == x.shape
    _chk(load().devqa_layernorm_bwd_dx(_p(x), _p(add), _p(gamma), _p(dy), M, D, float(eps), _p(dx), _stream()),
         "devqa_layernorm_bwd_dx")
    return dx


def layernorm_bwd_params(x, dy, eps, dgamma, dbeta, add=None, accumulate=True, rms=False):
    """dgamma (+)= sum_r dy * xhat(x [+ add]), dbeta (+)= sum_r dy over the rows of fp32 [M, D] (in place).  rms: LlamaRMSNorm's xhat
    (no mean subtraction), dbeta may be None."""
    M, D = x.shape
    _need(x, torch.float32, "x"); _need(dy, torch.float32, "dy"); _need(dgamma, torch.float32, "dgamma")
    assert dy.shape == x.shape and dgamma.numel() == D and (dbeta is None or dbeta.numel() == D) and (rms or dbeta is not None)
    if dbeta is not None:
        _need(dbeta, torch.float32, "dbeta")
    ws = torch.empty((2 * max(M, 1) + D,), dtype=torch.float32, device=x.device)
    _chk(load().devqa_layernorm_bwd_params(_p(x), _p(add), _p(dy), M, D, float(eps), int(bool(rms)), int(bool(accumulate)), _p(dgamma),
                                           _p(dbeta), _p(ws), _stream()), "devqa_layernorm_bwd_params")


def colsum_(x, out, accumulate=True):
    """out (+)= column sums of fp32 [M, D] (bias gradients)."""
    M, D = x.shape
    _need(x, torch.float32, "x"); _need(out, torch.float32, "out")
    assert out.numel() == D
    _chk(load().devqa_colsum_f32(_p(x), M, D, int(bool(accumulate)), _p(out), _stream()), "devqa_colsum_f32")


_ATT_ENV_KEYS = ("DEVQA_ATTENTION_DMA", "DEVQA_ATTENTION_NW", "DEVQA_ATTENTION_QB", "DEVQA_ATTENTION_DBUF", "DEVQA_ATTENTION_SHORT",
                 "DEVQA_ATTENTION_RESIDENT", "DEVQA_ATTENTION_EXP", "DEVQA_ATTENTION_XCD", "DEVQA_ATTENTION_RING", "DEVQA_ATTENTION_FOLD", "DEVQA_ATTENTION_NBUF", "DEVQA_ATTENTION_PACK")
_att_env_seen = None


def attention_env_sync():
    """The library reads its DEVQA_ATTENTION_* variant switches once; tests and A/B tools flip them between calls, so the op-level
    wrapper has them re-read when this process changed one (a handful of dict lookups per call)."""
    global _att_env_seen
    cur = tuple(os.environ.get(k) for k in _ATT_ENV_KEYS)
    if cur != _att_env_seen:
        if _att_env_seen is not None or any(v is not None for v in cur):
            _chk(load().devqa_attention_reload_env(), "devqa_attention_reload_env")
        _att_env_seen = cur


def attention(q, k, v, seq_desc, n_seq, max_q_len, H, dh, scale, causal, out=None, self_full=False):
    """q,k,v: bf16 2-D views (rows x >=H*dh, unit inner stride). seq_desc int32 [n_seq,6] on device.
    self_full=True promises plain non-causal self-attention for every sequence (kp_len == 0, ko_len == q_len): devqa.h, causal bit 2."""
    if self_full:
        assert not causal
        causal = 4
    for t in (q, k, v):
        assert t.dtype == q.dtype and t.is_cuda and t.dim() == 2 and t.stride(1) == 1
    assert q.dtype in (torch.bfloat16, torch.float32)
    assert seq_desc.dtype == torch.int32 and seq_desc.is_cuda and seq_desc.is_contiguous()
    if out is None:
        out = torch.empty((q.shape[0], H * dh), dtype=q.dtype, device=q.device)
    attention_env_sync()
    fn = load().devqa_attention if q.dtype == torch.bfloat16 else load().devqa_attention_f32
    _chk(fn(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0),
                                _p(seq_desc), int(n_seq), int(max_q_len), int(H), int(dh), float(scale), int(causal),
                                _stream()), "devqa_attention")
    return out


def im2col_patches(pixels, P, Kpad, dtype=torch.bfloat16):
    _need(pixels, torch.float32, "im2col pixels")
    B, C, S, S2 = pixels.shape
    assert C == 3 and S == S2
    out = torch.empty((B * (S // P) ** 2, Kpad), dtype=dtype, device=pixels.device)
    fn = load().devqa_im2col_patches if dtype == torch.bfloat16 else load().devqa_im2col_patches_f32
    _chk(fn(_p(pixels), B, S, P, Kpad, _p(out), _stream()), "devqa_im2col_patches")
    return out


def vit_assemble(patches, cls, pos, B, np_, D):
    _need(patches, torch.float32, "vit_assemble patches")
    out = torch.empty((B * (np_ + 1), D), dtype=torch.float32, device=patches.device)
    _chk(load().devqa_vit_assemble(_p(patches), _p(cls), _p(pos), B, np_, D, _p(out), _stream()), "devqa_vit_assemble")
    return out


def embed_rows(token, src_row, pos, embed, rows_f32, pos_table):
    R = token.numel()
    D = embed.shape[1]
    out = torch.empty((R, D), dtype=torch.float32, device=embed.device)
    n_rows = 0 if rows_f32 is None else rows_f32.shape[0]
    assert embed.dtype == pos_table.dtype and embed.is_contiguous() and pos_table.is_contiguous()
    fn = load().devqa_embed_rows if embed.dtype == torch.bfloat16 else load().devqa_embed_rows_f32
    _chk(fn(_p(token), _p(src_row), _p(pos), _p(embed), _p(rows_f32), _p(pos_table), R, D,
                                 embed.shape[0], n_rows, pos_table.shape[0], _p(out), _stream()), "devqa_embed_rows")
    return out


def gather_rows(x, idx):
    assert x.is_contiguous() and x.dim() == 2 and idx.dtype == torch.int32
    out = torch.empty((idx.numel(), x.shape[1]), dtype=x.dtype, device=x.device)
    _chk(load().devqa_gather_rows(_p(x), _p(idx), idx.numel(), x.shape[1], x.element_size(), _p(out), _stream()),
         "devqa_gather_rows")
    return out


def cast_f32_bf16(x, out=None):
    _need(x, torch.float32, "cast in")
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _chk(load().devqa_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "devqa_cast_f32_bf16")
    return out


def act_cast(x, act=ACT_RELU, want="bf16"):
    """act(x) of fp32 pre-activations -> bf16 (new tensor) or fp32 (in place)"""
    _need(x, torch.float32, "act_cast in")
    if want == "bf16":
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        _chk(load().devqa_act_cast(_p(x), int(act), _p(out), None, x.numel(), _stream()), "devqa_act_cast")
        return out
    _chk(load().devqa_act_cast(_p(x), int(act), None, _p(x), x.numel(), _stream()), "devqa_act_cast")
    return x


def vocab_rows(logits, labels=None, coef=None, want_argmax=True, want_nll=False, want_dlogits=False,
               dlogits_dtype=torch.bfloat16):
    _need(logits, torch.float32, "vocab_rows logits")
    R, V = logits.shape
    dev = logits.device
    am = torch.empty((R,), dtype=torch.int32, device=dev) if want_argmax else None
    nll = torch.empty((R,), dtype=torch.float32, device=dev) if want_nll else None
    dl = torch.empty((R, V), dtype=dlogits_dtype, device=dev) if want_dlogits else None
    fn = load().devqa_vocab_rows if dlogits_dtype == torch.bfloat16 else load().devqa_vocab_rows_f32
    _chk(fn(_p(logits), logits.stride(0), R, V, _p(labels), _p(coef), _p(am), _p(nll), _p(dl), V,
                                 _stream()), "devqa_vocab_rows")
    return am, nll, dl


def ft_adamw_step(w, m, v, w0, a, dy, y, do_update, adam_t, lr, beta1, beta2, eps, wd, clamp_eps):
    E, Dout, Din = w.shape
    Lmax = a.shape[1]
    for t, n in ((w, "w"), (m, "m"), (v, "v"), (w0, "w0"), (a, "a"), (dy, "dy"), (y, "y")):
        _need(t, torch.float32, "ft_adamw_step " + n)
    assert a.shape == (E, Lmax, Din) and dy.shape == (E, Lmax, Dout) and y.shape == (E, Lmax, Dout)
    assert w0.shape in ((Dout, Din), (E, Dout, Din)) and m.shape == w.shape and v.shape == w.shape
    assert do_update.dtype == torch.int32 and adam_t.dtype == torch.int32
    w0_stride = Dout * Din if w0.dim() == 3 else 0
    _chk(load().devqa_ft_adamw_step(_p(w), _p(m), _p(v), _p(w0), _p(a), _p(dy), _p(y), _p(do_update), _p(adam_t), E, Lmax,
                                    Dout, Din, float(lr), float(beta1), float(beta2), float(eps), float(wd),
                                    float(clamp_eps), w0_stride, _stream()), "devqa_ft_adamw_step")


def ft_adamw_step_fm(w, dstate, v, w0, a, dy, y, do_update, adam_t, lr, beta1, beta2, eps, wd, clamp_eps, single=None):
    """ft_adamw_step without a first-moment matrix (include/devqa.h, devqa_ft_adamw_step_fm): `dstate` fp32 [E, Lmax + 1, Dout] holds the EMA
    of dy (and, for edits flagged in `single` int32 [E] -- one loss row, in slot 0 -- the EMA of dy[0]^2: their second moment factors too and
    v is not touched); a[e] must not change between two first updates of edit e."""
    E, Dout, Din = w.shape
    Lmax = a.shape[1]
    for t, n in ((w, "w"), (dstate, "dstate"), (v, "v"), (w0, "w0"), (a, "a"), (dy, "dy"), (y, "y")):
        _need(t, torch.float32, "ft_adamw_step_fm " + n)
    assert a.shape == (E, Lmax, Din) and dy.shape == (E, Lmax, Dout) and y.shape == (E, Lmax, Dout)
    assert w0.shape in ((Dout, Din), (E, Dout, Din)) and dstate.shape == (E, Lmax + 1, Dout) and v.shape == w.shape
    assert do_update.dtype == torch.int32 and adam_t.dtype == torch.int32
    assert single is None or (single.dtype == torch.int32 and single.shape == (E,) and single.is_cuda)
    w0_stride = Dout * Din if w0.dim() == 3 else 0
    _chk(load().devqa_ft_adamw_step_fm(_p(w), _p(dstate), _p(v), _p(w0), _p(a), _p(dy), _p(y), _p(do_update), _p(adam_t), _p(single), E, Lmax,
                                       Dout, Din, float(lr), float(beta1), float(beta2), float(eps), float(wd),
                                       float(clamp_eps), w0_stride, _stream()), "devqa_ft_adamw_step_fm")


def active_columns(a):
    """a fp32 [E,L,Din] -> (idx int32 [E,Din] ascending active columns, count int32 [E])"""
    _need(a, torch.float32, "active_columns a")
    E, L, Din = a.shape
    idx = torch.empty((E, Din), dtype=torch.int32, device=a.device)
    cnt = torch.empty((E,), dtype=torch.int32, device=a.device)
    _chk(load().devqa_active_columns(_p(a), E, L, Din, _p(idx), _p(cnt), _stream()), "devqa_active_columns")
    return idx, cnt


def gather_cols(src, idx, count, npad, per_edit):
    """out[e,row,c] = src[(e,) row, idx[e,c]] for c < count[e] else 0.  src [rows,Din] (shared) or [E,rows,Din];
    idx int32 [E,Din'], count int32 [E]."""
    assert src.is_contiguous() and idx.dtype == torch.int32 and count.dtype == torch.int32
    E = idx.shape[0]
    rows, ld = src.shape[-2], src.shape[-1]
    out = torch.empty((E, rows, npad), dtype=src.dtype, device=src.device)
    fn = load().devqa_gather_cols_f32 if src.dtype == torch.float32 else load().devqa_gather_cols_bf16
    _chk(fn(_p(src), rows * ld if per_edit else 0, ld, rows, _p(idx), idx.stride(0), _p(count), E, int(npad), _p(out), _stream()),
         "devqa_gather_cols")
    return out


def scatter_cols_add(comp, idx_e, count_e, dense):
    """dense[:, idx_e[c]] += comp[:, c] for c < count_e (one edit)."""
    rows, npad = comp.shape
    _chk(load().devqa_scatter_cols_add_f32(_p(comp), rows, _p(idx_e), _p(count_e), npad, _p(dense), dense.stride(0),
                                           _stream()), "devqa_scatter_cols_add_f32")


def rows_matvec(w, a, bias=None, resid=None, shared=False):
    """y[e,l,:] = W_e . a[e,l,:] (+bias +resid). w fp32 [E,Dout,Din] or shared [Dout,Din]; a fp32 [E,L,Din]."""
    _need(w, torch.float32, "rows_matvec w")
    _need(a, torch.float32, "rows_matvec a")
    E, L, Din = a.shape
    Dout = w.shape[-2]
    assert w.shape[-1] == Din
    stride_e = 0 if (shared or w.dim() == 2) else Dout * Din
    y = torch.empty((E, L, Dout), dtype=torch.float32, device=a.device)
    if resid is not None:
        _need(resid, torch.float32, "rows_matvec resid")
        assert resid.shape == y.shape
    _chk(load().devqa_rows_matvec_f32(_p(w), stride_e, _p(a), _p(bias), _p(resid), _p(y), E, L, Dout, Din, _stream()),
         "devqa_rows_matvec_f32")
    return y


def delta_op(mode, w, w0=None, delta=None):
    _chk(load().devqa_delta_op(int(mode), _p(w), _p(w0), _p(delta), w.numel(), _stream()), "devqa_delta_op")


def ft_step_control(nll, mask, step, max_steps, floor, active, do_update, n_steps, adam_t, losses):
    E, Lmax = mask.shape
    _chk(load().devqa_ft_step_control(_p(nll), _p(mask), E, Lmax, int(step), int(max_steps), float(floor), _p(active),
                                      _p(do_update), _p(n_steps), _p(adam_t), _p(losses), _stream()),
         "devqa_ft_step_control")


def row_inv_norm(rows):
    """fp32 [R] = 1 / ||rows[r]|| (0 for a zero row): the corpus-side cache of cosine_topk(corpus_inv_norm=...)"""
    _need(rows, torch.float32, "row_inv_norm rows")
    out = torch.empty((rows.shape[0],), dtype=torch.float32, device=rows.device)
    _chk(load().devqa_row_inv_norm(_p(rows), rows.shape[0], rows.shape[1], _p(out), _stream()), "devqa_row_inv_norm")
    return out


def cosine_topk(corpus, queries, k, normalize_corpus=True, normalize_queries=True, corpus_inv_norm=None):
    _need(corpus, torch.float32, "cosine_topk corpus")
    _need(queries, torch.float32, "cosine_topk queries")
    N, D = corpus.shape
    Q = queries.shape[0]
    ws_bytes = load().devqa_cosine_topk_workspace(N, Q, k)
    ws = torch.empty((ws_bytes + 256,), dtype=torch.uint8, device=corpus.device)
    off = (-ws.data_ptr()) % 256
    idx = torch.empty((Q, k), dtype=torch.int64, device=corpus.device)
    sc = torch.empty((Q, k), dtype=torch.float32, device=corpus.device)
    if corpus_inv_norm is not None and normalize_corpus:
        _need(corpus_inv_norm, torch.float32, "cosine_topk corpus_inv_norm")
        assert corpus_inv_norm.numel() == N
        _chk(load().devqa_cosine_topk_cached(_p(corpus), _p(corpus_inv_norm), _p(queries), N, Q, D, int(k), int(normalize_queries),
                                             _p(idx), _p(sc), c_void_p(ws.data_ptr() + off), _stream()), "devqa_cosine_topk_cached")
        return idx, sc
    _chk(load().devqa_cosine_topk(_p(corpus), _p(queries), N, Q, D, int(k), int(normalize_corpus), int(normalize_queries),
                                  _p(idx), _p(sc), c_void_p(ws.data_ptr() + off), _stream()), "devqa_cosine_topk")
    return idx, sc


def attention_bwd(q, k, v, o, d_out, seq_desc, n_seq, max_len, H, dh, scale, causal):
    """Gradients (dq, dk, dv) of `attention` for descriptors without a visible prefix (kp_len == 0)."""
    for t in (q, k, v, o, d_out):
        assert t.dtype == q.dtype and t.is_cuda and t.dim() == 2 and t.stride(1) == 1
    assert q.dtype in (torch.bfloat16, torch.float32)
    R = q.shape[0]
    dq = torch.zeros((R, H * dh), dtype=q.dtype, device=q.device)
    dk, dv = (torch.zeros((k.shape[0], H * dh), dtype=q.dtype, device=q.device) for _ in range(2))     # (cross-attention: keys are other rows)
    stats = torch.empty((R * H * 2,), dtype=torch.float32, device=q.device)
    fn = load().devqa_attention_bwd if q.dtype == torch.bfloat16 else load().devqa_attention_bwd_f32
    _chk(fn(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0), _p(d_out), d_out.stride(0),
            _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv), dv.stride(0), _p(stats), _p(seq_desc), int(n_seq), int(max_len),
            int(H), int(dh), float(scale), int(causal), _stream()), "devqa_attention_bwd")
    return dq, dk, dv


def relu_bwd(act_out, grad_out):
    assert act_out.dtype == grad_out.dtype and act_out.shape == grad_out.shape and act_out.is_contiguous() and grad_out.is_contiguous()
    out = torch.empty_like(grad_out)
    fn = load().devqa_relu_bwd if act_out.dtype == torch.bfloat16 else load().devqa_relu_bwd_f32
    _chk(fn(_p(act_out), _p(grad_out), _p(out), act_out.numel(), _stream()), "devqa_relu_bwd")
    return out


def gelu(x, want="bf16"):
    """HF "gelu" of fp32 pre-activations (kept by the caller for gelu_bwd) -> operand dtype"""
    _need(x, torch.float32, "gelu x")
    ob = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want == "bf16" else None
    of = torch.empty_like(x) if want == "f32" else None
    _chk(load().devqa_gelu_f32(_p(x), _p(ob), _p(of), x.numel(), _stream()), "devqa_gelu_f32")
    return ob if ob is not None else of


def gelu_bwd(x, grad_out):
    _need(x, torch.float32, "gelu_bwd x")
    _need(grad_out, torch.float32, "gelu_bwd grad_out")
    assert x.shape == grad_out.shape
    out = torch.empty_like(x)
    _chk(load().devqa_gelu_bwd_f32(_p(x), _p(grad_out), _p(out), x.numel(), _stream()), "devqa_gelu_bwd_f32")
    return out


def mend_normalize_concat(u, v, idx=None, u_mean=None, u_std=None, v_mean=None, v_std=None, eps=1e-7):
    _need(u, torch.float32, "mend_normalize_concat u")
    _need(v, torch.float32, "mend_normalize_concat v")
    n = u.shape[0] if idx is None else idx.numel()
    out = torch.empty((n, u.shape[1] + v.shape[1]), dtype=torch.float32, device=u.device)
    if n:
        _chk(load().devqa_mend_normalize_concat(_p(u), _p(v), _p(idx), _p(u_mean), _p(u_std), _p(v_mean), _p(v_std), float(eps), n,
                                                u.shape[1], v.shape[1], _p(out), _stream()), "devqa_mend_normalize_concat")
    return out


def mend_lrlinear_epilogue(pre, bias, scale, shift, x):
    _need(pre, torch.float32, "mend_lrlinear_epilogue pre")
    _need(x, torch.float32, "mend_lrlinear_epilogue x")
    out = torch.empty_like(pre)
    _chk(load().devqa_mend_lrlinear_epilogue(_p(pre), _p(bias), _p(scale), _p(shift), _p(x), _p(out), pre.shape[0], pre.shape[1],
                                             _stream()), "devqa_mend_lrlinear_epilogue")
    return out


def logit_kl_rows(l1, l2):
    """fp32 [R]: KL(softmax(l1[r]) || softmax(l2[r])) per row (2-D fp32 views, unit inner stride)."""
    assert l1.dtype == l2.dtype == torch.float32 and l1.shape == l2.shape and l1.stride(1) == 1 and l2.stride(1) == 1
    kl = torch.empty((l1.shape[0],), dtype=torch.float32, device=l1.device)
    _chk(load().devqa_logit_kl_rows(_p(l1), l1.stride(0), _p(l2), l2.stride(0), l1.shape[0], l1.shape[1], _p(kl), _stream()),
         "devqa_logit_kl_rows")
    return kl


def kl_dlogits(l1, l2, coef, dtype=torch.float32):
    """-> (kl [R] fp32, dlogits2 [R,V] `dtype`): rows of KL(softmax(l1) || softmax(l2)) and coef * (softmax(l2) - softmax(l1))."""
    assert l1.dtype == l2.dtype == torch.float32 and l1.shape == l2.shape and l1.stride(1) == 1 and l2.stride(1) == 1
    _need(coef, torch.float32, "kl_dlogits coef")
    R, V = l1.shape
    kl = torch.empty((R,), dtype=torch.float32, device=l1.device)
    d = torch.empty((R, V), dtype=dtype, device=l1.device)
    _chk(load().devqa_kl_dlogits(_p(l1), l1.stride(0), _p(l2), l2.stride(0), R, V, _p(coef), _p(kl), _p(d), V,
                                 int(dtype == torch.bfloat16), _stream()), "devqa_kl_dlogits")
    return kl, d


def welford_rows(x, idx, reset, mean, s, std, k):
    """In-place update of mean / s / std; returns the NEW sample counter tensor (k itself is left untouched)."""
    _need(x, torch.float32, "welford_rows x")
    n = x.shape[0] if idx is None else idx.numel()
    k_out = torch.empty_like(k)
    _chk(load().devqa_welford_rows(_p(x), _p(idx), n, x.shape[1], int(bool(reset)), _p(mean), _p(s), _p(std), _p(k), _p(k_out),
                                   _stream()), "devqa_welford_rows")
    return k_out


def mend_lrlinear_bwd(pre, bias, scale, shift, dout, g_scale, g_shift, g_bias):
    _need(pre, torch.float32, "mend_lrlinear_bwd pre")
    _need(dout, torch.float32, "mend_lrlinear_bwd dout")
    dpre = torch.empty_like(pre)
    _chk(load().devqa_mend_lrlinear_bwd(_p(pre), _p(bias), _p(scale), _p(shift), _p(dout), pre.shape[0], pre.shape[1], _p(dpre),
                                        _p(g_scale), _p(g_shift), _p(g_bias), _stream()), "devqa_mend_lrlinear_bwd")
    return dpre


def sumsq_(x, out):
    _need(x, torch.float32, "sumsq x")
    _chk(load().devqa_sumsq_f32(_p(x), x.numel(), _p(out), _stream()), "devqa_sumsq_f32")


def adam_step_(p, grad, m, v, lr, step, grad_scale=None, beta1=0.9, beta2=0.999, eps=1e-8):
    for t in (p, grad, m, v):
        _need(t, torch.float32, "adam_step tensor")
    _chk(load().devqa_adam_step(_p(p), _p(grad), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), int(step),
                                _p(grad_scale), _stream()), "devqa_adam_step")


def tp_neuron_fwd(h, k, b, lab, v, ybase):
    for t in (h, k, b, v, ybase):
        _need(t, torch.float32, "tp_neuron_fwd tensor")
    T, d = h.shape
    L, d_out = ybase.shape
    pre = torch.empty((T,), dtype=torch.float32, device=h.device)
    y = torch.empty_like(ybase)
    _chk(load().devqa_tp_neuron_fwd(_p(h), T, d, _p(k), _p(b), _p(lab), L, _p(v), _p(ybase), d_out, _p(pre), _p(y), _stream()),
         "devqa_tp_neuron_fwd")
    return pre, y


def tp_neuron_bwd(h, pre, lab, dy, hm, k, b, v, lambda_a, lambda_m, weight_decay):
    for t in (h, pre, dy, hm, k, b, v):
        _need(t, torch.float32, "tp_neuron_bwd tensor")
    T, d = h.shape
    L, d_out = dy.shape
    Tm = hm.shape[0]
    scratch = torch.empty((T + Tm,), dtype=torch.float32, device=h.device)
    gk, gb, gv = torch.empty_like(k), torch.empty_like(b), torch.empty_like(v)
    losses = torch.empty((2,), dtype=torch.float32, device=h.device)
    _chk(load().devqa_tp_neuron_bwd(_p(h), _p(pre), T, d, _p(lab), L, _p(dy), d_out, _p(hm), Tm, _p(k), _p(b), _p(v), float(lambda_a),
                                    float(lambda_m), float(weight_decay), _p(scratch), _p(gk), _p(gb), _p(gv), _p(losses), _stream()),
         "devqa_tp_neuron_bwd")
    return gk, gb, gv, losses


def tp_gated_neuron_fwd(h, K2, B2, lab, v, ybase):
    """LLaMA-FFN patch neuron: K2 [2,d] (gate key, up key), B2 [2] -> pre [2,T], y [L,d_out]"""
    for t in (h, K2, B2, v, ybase):
        _need(t, torch.float32, "tp_gated_neuron_fwd tensor")
    T, d = h.shape
    assert tuple(K2.shape) == (2, d) and B2.numel() == 2
    L, d_out = ybase.shape
    pre = torch.empty((2, T), dtype=torch.float32, device=h.device)
    y = torch.empty_like(ybase)
    _chk(load().devqa_tp_gated_neuron_fwd(_p(h), T, d, _p(K2), _p(B2), _p(lab), L, _p(v), _p(ybase), d_out, _p(pre), _p(y), _stream()),
         "devqa_tp_gated_neuron_fwd")
    return pre, y


def tp_gated_neuron_bwd(h, pre, lab, dy, hm, K2, B2, v, lambda_a, lambda_m, weight_decay):
    for t in (h, pre, dy, hm, K2, B2, v):
        _need(t, torch.float32, "tp_gated_neuron_bwd tensor")
    T, d = h.shape
    assert tuple(K2.shape) == (2, d) and tuple(pre.shape) == (2, T)
    L, d_out = dy.shape
    Tm = hm.shape[0]
    scratch = torch.empty((2 * (T + Tm),), dtype=torch.float32, device=h.device)
    gk, gb, gv = torch.empty_like(K2), torch.empty_like(B2), torch.empty_like(v)
    losses = torch.empty((2,), dtype=torch.float32, device=h.device)
    _chk(load().devqa_tp_gated_neuron_bwd(_p(h), _p(pre), T, d, _p(lab), L, _p(dy), d_out, _p(hm), Tm, _p(K2), _p(B2), _p(v),
                                          float(lambda_a), float(lambda_m), float(weight_decay), _p(scratch), _p(gk), _p(gb), _p(gv),
                                          _p(losses), _stream()), "devqa_tp_gated_neuron_bwd")
    return gk, gb, gv, losses


def swiglu_bwd(gu, da):
    """gu [R,2F] (gate | up; bf16 or fp32), da fp32 [R,F] -> dgu fp32 [R,2F]"""
    assert gu.dim() == 2 and gu.is_contiguous() and gu.shape[1] % 2 == 0
    _need(da, torch.float32, "swiglu_bwd da")
    R, F2 = gu.shape
    out = torch.empty((R, F2), dtype=torch.float32, device=gu.device)
    fn = load().devqa_swiglu_bwd_bf16 if gu.dtype == torch.bfloat16 else load().devqa_swiglu_bwd_f32
    _chk(fn(_p(gu), _p(da), R, F2 // 2, _p(out), _stream()), "devqa_swiglu_bwd")
    return out


# ---------------------------------------------------------------------------------------------
# path level: model context + the launch schedules behind the ABI (include/devqa.h "PATH LEVEL")
# ---------------------------------------------------------------------------------------------
DTYPE_BF16, DTYPE_F32, FAMILY_BLIP2_OPT, SCORE_COLS = 1, 2, 1, 16
FAMILY_LLAVA, FAMILY_MINIGPT4 = 2, 3


class ModelDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("family", "compute_dtype", "image_size", "patch_size", "v_hidden", "v_layers", "v_heads", "v_ffn",
                                             "q_hidden", "q_layers", "q_heads", "q_ffn", "q_cross_freq", "num_query_tokens",
                                             "t_hidden", "t_layers", "t_heads", "t_ffn", "t_vocab", "t_max_pos")] + \
               [(n, ctypes.c_float) for n in ("v_ln_eps", "q_ln_eps", "t_ln_eps", "t_rms_eps", "t_rope_theta")] + \
               [(n, ctypes.c_int32) for n in ("v_run_layers", "t_flags")]


class WeightEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("ptr", c_void_p), ("dtype", ctypes.c_int32), ("ndim", ctypes.c_int32), ("shape", c_int64 * 4)]


MEND_MAX_LAYERS = 4
FT_MAX_ROWS = 64          # DEVQA_FT_MAX_ROWS (include/devqa.h): loss rows per edit in the FT_VL loop


class MendLayer(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("u", "v", "bias", "mode_scale", "mode_shift")]


class MendNet(ctypes.Structure):
    _fields_ = [("n_layers", ctypes.c_int32), ("rank", ctypes.c_int32), ("u_mean", c_void_p), ("u_std", c_void_p), ("v_mean", c_void_p),
                ("v_std", c_void_p), ("layers", MendLayer * MEND_MAX_LAYERS), ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32)]


MEND_SPLIT_BF16 = 1


def mend_transform(x, delta, idx, layers, stats=None, split_bf16=False):
    """K16 (devqa_mend_transform): rows `idx` (int32, or None = all) of fp32 x [R, du] / delta [R, dv] through the hyper-network in
    inference mode.  layers: list of dicts {u [D, rank], v [rank, D], bias, mode_scale, mode_shift [D]} (fp32, contiguous; the mode row
    of the edited module already selected); stats: (u_mean, u_std, v_mean, v_std) or None.  -> (x~ [n, du], delta~ [n, dv])"""
    _need(x, torch.float32, "mend_transform x"); _need(delta, torch.float32, "mend_transform delta")
    du, dv = x.shape[1], delta.shape[1]
    n = x.shape[0] if idx is None else int(idx.numel())
    out_x = torch.empty((n, du), dtype=torch.float32, device=x.device)
    out_d = torch.empty((n, dv), dtype=torch.float32, device=x.device)
    if n == 0:
        return out_x, out_d
    net = MendNet()
    net.n_layers, net.rank = len(layers), int(layers[0]["v"].shape[0])
    net.flags = MEND_SPLIT_BF16 if split_bf16 else 0     # the GEMMs as three bf16 MFMA products of split operands (bf16 compute mode)
    keep = []
    if stats is not None:
        for name, t in zip(("u_mean", "u_std", "v_mean", "v_std"), stats):
            _need(t, torch.float32, "mend_transform " + name)
            setattr(net, name, t.data_ptr())
    for i, L in enumerate(layers):
        for name in ("u", "v", "bias", "mode_scale", "mode_shift"):
            t = L[name]
            _need(t, torch.float32, "mend_transform layer %d %s" % (i, name))
            keep.append(t)
            setattr(net.layers[i], name, t.data_ptr())
    nb = load().devqa_mend_transform_workspace(n, du, dv, net.rank)
    kws, ws = _ws(nb, x.device)
    _chk(load().devqa_mend_transform(_p(x), _p(delta), _p(idx), n, du, dv, ctypes.byref(net), _p(out_x), _p(out_d), ws, nb, _stream()),
         "devqa_mend_transform")
    return out_x, out_d


def mend_apply_(h, xt, dtT, y):
    """K17 (devqa_mend_apply): y [R, dout] fp32 += (h [R, din] . xt^T) . dt for the low-rank factors xt [npad, din], dtT [dout, npad]
    (operands in h's dtype: bf16 or fp32; npad % 64 == 0)."""
    assert h.dtype == xt.dtype == dtT.dtype and h.dtype in (torch.bfloat16, torch.float32)
    for t in (h, xt, dtT):
        assert t.is_cuda and t.is_contiguous()
    _need(y, torch.float32, "mend_apply y")
    R, din = h.shape
    npad, dout = xt.shape[0], dtT.shape[0]
    assert xt.shape[1] == din and dtT.shape[1] == npad and y.shape == (R, dout)
    cd = 2 if h.dtype == torch.float32 else 1
    nb = load().devqa_mend_apply_workspace(R, npad, cd)
    kws, ws = _ws(nb, h.device)
    _chk(load().devqa_mend_apply(_p(h), _p(xt), _p(dtT), _p(y), R, din, dout, npad, cd, ws, nb, _stream()), "devqa_mend_apply")
    return y


class FtCfg(ctypes.Structure):
    _fields_ = [("num_steps", ctypes.c_int32), ("lr", c_float), ("weight_decay", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("loss_floor", c_float), ("clamp_eps", c_float)]


def _ws(nbytes, device):
    """256-byte aligned scratch of `nbytes` from torch's caching allocator -> (keep-alive tensor, pointer)"""
    if nbytes < 0:
        raise DevqaError("workspace query failed (bad context handle or dims)")
    buf = torch.empty((nbytes + 256,), dtype=torch.uint8, device=device)
    return buf, c_void_p(buf.data_ptr() + (-buf.data_ptr()) % 256)


class PathContext:
    """Owner of one devqa_ctx_t: the table of named device tensors stays referenced here for the context's lifetime."""

    def __init__(self, device_index, desc: ModelDesc, weights: dict):
        self._names = [n.encode() for n in weights]
        self.weights = dict(weights)
        arr = (WeightEntry * len(weights))()
        for i, (n, t) in enumerate(weights.items()):
            if not t.is_cuda or not t.is_contiguous():
                raise DevqaError("weight %s must be a contiguous device tensor" % n)
            dt = {torch.bfloat16: DTYPE_BF16, torch.float32: DTYPE_F32}.get(t.dtype)
            if dt is None or t.dim() < 1 or t.dim() > 4:
                raise DevqaError("weight %s: unsupported dtype / rank" % n)
            arr[i].name, arr[i].ptr, arr[i].dtype, arr[i].ndim = self._names[i], t.data_ptr(), dt, t.dim()
            for k in range(t.dim()):
                arr[i].shape[k] = t.shape[k]
        h = ctypes.c_uint64()
        _chk(load().devqa_ctx_create(int(device_index), ctypes.byref(desc), arr, len(weights), ctypes.byref(h)), "devqa_ctx_create")
        self.h = h.value
        self.desc = desc
        self.device = torch.device("cuda", int(device_index))
        self.adt = torch.bfloat16 if desc.compute_dtype == DTYPE_BF16 else torch.float32

    def set_weight(self, name, t):
        self.weights[name] = t
        _chk(load().devqa_ctx_set_weight(self.h, name.encode(), _p(t)), "devqa_ctx_set_weight")

    def close(self):
        if getattr(self, "h", None):
            _chk(load().devqa_ctx_destroy(self.h), "devqa_ctx_destroy")
            self.h = None

    def __del__(self):
        try:
            if getattr(self, "h", None) and _lib is not None:
                _lib.devqa_ctx_destroy(self.h)
        except Exception:
            pass

    # ---- schedules ----
    def vision_encode(self, pixels):
        _need(pixels, torch.float32, "vision_encode pixels")
        B, d = pixels.shape[0], self.desc
        n_tok = (d.image_size // d.patch_size) ** 2 if d.family == FAMILY_LLAVA else d.num_query_tokens
        out = torch.empty((B, n_tok, d.t_hidden), dtype=torch.float32, device=pixels.device)
        n = load().devqa_vision_encode_workspace(self.h, B)
        keep, ws = _ws(n, pixels.device)
        _chk(load().devqa_vision_encode(self.h, _p(pixels), B, _p(out), ws, n, _stream()), "devqa_vision_encode")
        return out

    def llm_layers(self, x, seq_desc, n_seq, max_len, dense, n_layers=-1, stop_before_fc2=False, positions=None, first_layer=0):
        """Layers [first_layer, first_layer + n_layers) in place on x fp32 [R, d] (devqa_llm_layers_ex; `positions` int32 [R]: the rows'
        rotary positions, required by the LLaMA-family decoders).  -> fc2 / down_proj input [R, ffn] (compute dtype) when
        stop_before_fc2, else None"""
        _need(x, torch.float32, "llm_layers x")
        R = x.shape[0]
        if positions is not None:
            assert positions.dtype == torch.int32 and positions.is_cuda and positions.is_contiguous() and positions.numel() == R
        a = torch.empty((R, self.desc.t_ffn), dtype=self.adt, device=x.device) if stop_before_fc2 else None
        n = load().devqa_llm_layers_workspace(self.h, R, int(stop_before_fc2))
        keep, ws = _ws(n, x.device)
        _chk(load().devqa_llm_layers_ex(self.h, _p(x), _p(positions), _p(seq_desc), int(n_seq), int(max_len), R, int(bool(dense)), int(first_layer),
                                        int(n_layers), int(bool(stop_before_fc2)), _p(a), ws, n, _stream()), "devqa_llm_layers_ex")
        return a

    def llm_prefix(self, x, seq_desc, n_seq, max_len, dense):
        """SURVEY 8(b)'s devqa_llm_prefix: all layers in place on x, the last one stopping at its fc2 input -> [R, ffn]"""
        _need(x, torch.float32, "llm_prefix x")
        R = x.shape[0]
        a = torch.empty((R, self.desc.t_ffn), dtype=self.adt, device=x.device)
        n = load().devqa_llm_prefix_workspace(self.h, R)
        keep, ws = _ws(n, x.device)
        _chk(load().devqa_llm_prefix(self.h, _p(x), _p(seq_desc), int(n_seq), int(max_len), R, int(bool(dense)), _p(a), ws, n, _stream()),
             "devqa_llm_prefix")
        return a

    def llm_head(self, rows, add=None):
        _need(rows, torch.float32, "llm_head rows")
        if add is not None:
            _need(add, torch.float32, "llm_head add")
            assert add.shape == rows.shape
        R = rows.shape[0]
        out = torch.empty((R, self.desc.t_vocab), dtype=torch.float32, device=rows.device)
        n = load().devqa_llm_head_workspace(self.h, R)
        keep, ws = _ws(n, rows.device)
        _chk(load().devqa_llm_head(self.h, _p(rows), _p(add), R, _p(out), ws, n, _stream()), "devqa_llm_head")
        return out

    def llm_forward(self, x, seq_desc, n_seq, max_len, dense, want_rows, positions=None):
        _need(x, torch.float32, "llm_forward x")
        assert want_rows.dtype == torch.int32 and want_rows.is_cuda
        R, Rw = x.shape[0], want_rows.numel()
        out = torch.empty((Rw, self.desc.t_vocab), dtype=torch.float32, device=x.device)
        n = load().devqa_llm_forward_workspace(self.h, R, Rw)
        keep, ws = _ws(n, x.device)
        _chk(load().devqa_llm_forward_ex(self.h, _p(x), _p(positions), _p(seq_desc), int(n_seq), int(max_len), R, int(bool(dense)), _p(want_rows), Rw,
                                         _p(out), ws, n, _stream()), "devqa_llm_forward_ex")
        return out

    def ft_edit(self, w0, a_rows, resid_rows, labels, mask, num_steps, lr, weight_decay, clamp_eps, beta1=0.9, beta2=0.999, eps=1e-8,
                loss_floor=1e-2):
        """-> (delta [E, d, npad], losses [E, num_steps], steps int32 [E], updates int32 [E]); see devqa_ft_edit"""
        for t, nm in ((w0, "w0"), (a_rows, "a_rows"), (resid_rows, "resid_rows"), (mask, "mask")):
            _need(t, torch.float32, "ft_edit " + nm)
        assert labels.dtype == torch.int32 and labels.is_cuda
        E, kmax, npad = a_rows.shape
        d = self.desc.t_hidden
        assert w0.shape in ((d, npad), (E, d, npad)) and tuple(resid_rows.shape) == (E * kmax, d) and tuple(mask.shape) == (E, kmax)
        dev = a_rows.device
        delta = torch.empty((E, d, npad), dtype=torch.float32, device=dev)
        losses = torch.empty((E, num_steps), dtype=torch.float32, device=dev)
        steps = torch.empty((E,), dtype=torch.int32, device=dev)
        updates = torch.empty((E,), dtype=torch.int32, device=dev)
        cfg = FtCfg(int(num_steps), float(lr), float(weight_decay), float(beta1), float(beta2), float(eps), float(loss_floor), float(clamp_eps))
        n = load().devqa_ft_edit_workspace(self.h, E, kmax, npad)
        keep, ws = _ws(n, dev)
        _chk(load().devqa_ft_edit(self.h, _p(w0), d * npad if w0.dim() == 3 else 0, _p(a_rows), _p(resid_rows), _p(labels), _p(mask), E, kmax,
                                  npad, ctypes.byref(cfg), _p(delta), _p(losses), _p(steps), _p(updates), ws, n, _stream()), "devqa_ft_edit")
        return delta, losses, steps, updates

    def bind_edit_target(self, name):
        _chk(load().devqa_ctx_bind_edit_target(self.h, name.encode(), _stream()), "devqa_ctx_bind_edit_target")

    def apply_delta(self, delta):
        _need(delta, torch.float32, "apply_delta delta")
        _chk(load().devqa_apply_delta(self.h, _p(delta), _stream()), "devqa_apply_delta")

    def restore(self):
        _chk(load().devqa_restore(self.h, _stream()), "devqa_restore")


def token_acc(logits_rows, labels, mask):
    """-> (acc fp32 [1], pred int32 [R]) on the device (devqa_token_acc)"""
    assert logits_rows.dtype == torch.float32 and logits_rows.dim() == 2 and logits_rows.stride(1) == 1
    R, V = logits_rows.shape
    assert labels.dtype == torch.int32 and labels.numel() == R
    _need(mask, torch.float32, "token_acc mask")
    acc = torch.empty((1,), dtype=torch.float32, device=logits_rows.device)
    pred = torch.empty((R,), dtype=torch.int32, device=logits_rows.device)
    _chk(load().devqa_token_acc(_p(logits_rows), logits_rows.stride(0), R, V, _p(labels), _p(mask), _p(acc), _p(pred), _stream()),
         "devqa_token_acc")
    return acc, pred


def comm_unique_id() -> bytes:
    buf = ctypes.create_string_buffer(128)
    _chk(load().devqa_comm_unique_id(buf), "devqa_comm_unique_id")
    return buf.raw


class ScoreComm:
    """RCCL communicator behind the ABI for the single gather of score rows."""

    def __init__(self, rank, world, uid: bytes, device_index):
        h = ctypes.c_uint64()
        _chk(load().devqa_comm_create(int(rank), int(world), ctypes.create_string_buffer(uid, 128), int(device_index), ctypes.byref(h)),
             "devqa_comm_create")
        self.h, self.rank, self.world = h.value, rank, world

    def gather_scores(self, local):
        """local fp32 [n, 16] (same n on every rank) -> fp32 [world * n, 16] on every rank"""
        _need(local, torch.float32, "gather_scores local")
        assert local.dim() == 2 and local.shape[1] == SCORE_COLS
        out = torch.empty((self.world * local.shape[0], SCORE_COLS), dtype=torch.float32, device=local.device)
        _chk(load().devqa_gather_scores(self.h, _p(local), local.shape[0], _p(out), _stream()), "devqa_gather_scores")
        return out

    def close(self):
        if self.h:
            _chk(load().devqa_comm_destroy(self.h), "devqa_comm_destroy")
            self.h = None
