"""ctypes binding of ``csrc/librpb_hip.so`` (C ABI declared in ``include/rpb.h``).

This is the stub a maintainer of the reference would add to call the MI355X path.
Loading is lazy; a missing library is a hard error -- there is no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 2          # RPB_ABI_VERSION of include/rpb.h
LIB_PATH = os.environ.get("RPB_LIB_PATH") or os.path.join(_HERE, "csrc", "librpb_hip.so")     # RPB_LIB_PATH: an instrumented build (tools/dbg)

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double
_T = {"p": _P, "i": _I, "l": _L, "f": _F, "d": _D}

# name -> (restype, argument codes)   p=pointer i=int l=long f=float d=double
SIGNATURES = {
    "rpb_last_error": (ctypes.c_char_p, ""),
    "rpb_abi_version": (_I, ""),
    "rpb_bf16_const_planes": (_I, ""),
    "rpb_lift_pad_fwd": (_I, "ppppppp" + "iiiiiiiii" + "p"),
    "rpb_lift_pad_fwd_bf16": (_I, "ppppppp" + "iiiiiiiii" + "p"),
    "rpb_axis_gemm_bf16in": (_I, "ppp" + "iiii" + "llll" + "i" + "p"),
    "rpb_cell_mix_bf16": (_I, "pppppp" + "l" + "iii" + "ppppi" + "i" + "p"),
    "rpb_axis_gemm_bf16out": (_I, "ppp" + "iiii" + "llll" + "i" + "p"),
    "rpb_proj_fwd_bf16": (_I, "pppppp" + "l" + "ii" + "iiiiii" + "i" + "p"),
    "rpb_spectrum_bin": (_I, "ppiip"),
    "rpb_feat_mix": (_I, "pppp" + "iiiii" + "p"),
    "rpb_feat_mix_wgrad_rows": (_I, ""),
    "rpb_feat_mix_wgrad": (_I, "ppp" + "iiiii" + "p"),
    "rpb_small_gemm": (_I, "ppp" + "iiiiiiiii" + "p"),
    "rpb_lift_feat": (_I, "ppppp" + "iiiiiiiii" + "p"),
    "rpb_cell_mix_feat": (_I, "ppppppp" + "l" + "iii" + "ppppi" + "p"),
    "rpb_bn_bwd_row_feat": (_I, "ppppppppp" + "d" + "i" + "ppp" + "iiiii" + "p"),
    "rpb_proj_bwd_fused_supported": (_I, "iiii"),
    "rpb_proj_dgrad_slots": (_L, "iii"),
    "rpb_proj_dgrad": (_I, "pppppppp" + "ii" + "iiiiii" + "ppppi" + "i" + "p"),
    "rpb_proj_wgrad_slots": (_L, "iii"),
    "rpb_proj_wgrad_row": (_I, "i"),
    "rpb_proj_wgrad_roles": (_I, ""),
    "rpb_proj_wgrad": (_I, "pppppp" + "ii" + "iiiiii" + "ppppi" + "i" + "p"),
    "rpb_stream_probe": (_I, "pppp" + "l" + "ii" + "p"),
    "rpb_mfma_probe": (_I, "pp" + "ii" + "p" + "p"),
    "rpb_head_bwd_supported": (_I, "iiiiii"),
    "rpb_head_bwd_slots": (_L, "iii"),
    "rpb_head_bwd_row": (_I, "i"),
    "rpb_head_bwd": (_I, "ppppppp" + "ii" + "iiiiii" + "pppp" + "p"),
    "rpb_head_bwd_finalize": (_I, "pppp" + "i" + "ppppp" + "p"),
    "rpb_head_fwd_bwd": (_I, "pppppp" + "f" + "ppp" + "ii" + "iiiiii" + "pppp" + "p"),
    "rpb_cell_mix_eval_dft_bf16": (_I, "pppppp" + "l" + "ii" + "ppppi" + "pip" + "pip"),
    "rpb_cell_mix_eval_crop": (_I, "pppppp" + "iiiiiiii" + "ppppi" + "iip"),
    "rpb_cell_mix_eval_dft_supported": (_I, "liii"),
    "rpb_cell_mix_eval_dft": (_I, "pppppp" + "l" + "iii" + "ppppi" + "pip" + "pp"),
    "rpb_cell_mix_eval_dft_f16x2": (_I, "pppppp" + "l" + "iii" + "ppppi" + "pip" + "p" + "i" + "p"),
    "rpb_cell_mix_eval_crop_c128": (_I, "pppppp" + "iiiiiiii" + "ppppi" + "p"),
    "rpb_cell_mix_eval_crop_c128_supported": (_I, "lii"),
    "rpb_proj_fwd_f16x2": (_I, "pppppp" + "l" + "iiiiiii" + "p"),
    "rpb_cell_mix_eval_crop_f16x2": (_I, "pppppp" + "iiiiiiii" + "ppppi" + "i" + "p"),
    "rpb_dpot_patch_tokens": (_I, "ppppp" + "iiiiiii" + "p"),
    "rpb_dpot_patch_tokens_bwd": (_I, "pp" + "iiiiiii" + "p"),
    "rpb_rowtable_add": (_I, "pp" + "l" + "iii" + "p"),
    "rpb_rowtable_grad": (_I, "pp" + "iiii" + "p"),
    "rpb_dpot_tagg_prep": (_I, "pppppp" + "ii" + "p"),
    "rpb_dpot_tagg_finish": (_I, "ppppppp" + "ii" + "p"),
    "rpb_gn_tokens_fwd": (_I, "pppppp" + "iiii" + "f" + "p"),
    "rpb_gn_tokens_bwd": (_I, "ppppppppp" + "iiii" + "p"),
    "rpb_afno_wprep": (_I, "pp" + "iiii" + "p"),
    "rpb_afno_mlp": (_I, "pppppppp" + "l" + "iii" + "p"),
    "rpb_afno_wgrad_splits": (_I, "l"),
    "rpb_afno_wgrad": (_I, "pppp" + "l" + "iii" + "p"),
    "rpb_dpot_unpatch": (_I, "pp" + "iiiiiiii" + "p"),
    "rpb_dpot_unpatch_bwd": (_I, "pp" + "iiiiiiii" + "p"),
    "rpb_dp_available": (_I, ""),
    "rpb_dp_unique_id": (_I, "p"),
    "rpb_dp_allreduce_init": (_I, "piip"),
    "rpb_dp_allreduce_enqueue": (_I, "pplip"),
    "rpb_dp_allreduce_wait": (_I, "pp"),
    "rpb_dp_allreduce_inline": (_I, "pplip"),
    "rpb_dp_allreduce_destroy": (_I, "p"),
    "rpb_dp_allreduce_abort": (_I, "p"),
    "rpb_dp_reduce_scatter_enqueue": (_I, "pplip"),
    "rpb_dp_allgather_enqueue": (_I, "pplip"),
    "rpb_dp_mark": (_I, "pi"),
    "rpb_dp_wait_mark": (_I, "pip"),
    "rpb_dp_set_model": (_I, "piff"),
    "rpb_dp_set_timing": (_I, "pi"),
    "rpb_dp_step_times": (_I, "ppi"),
    "rpb_dp_p2p_init": (_I, "iipppp" + "li" + "p"),
    "rpb_dp_p2p_slice": (_I, "ppp"),
    "rpb_dp_p2p_signal": (_I, "pilp"),
    "rpb_dp_p2p_wait": (_I, "pilp"),
    "rpb_dp_p2p_adam": (_I, "ppp" + "ffff" + "l" + "f" + "p"),
    "rpb_dp_p2p_destroy": (_I, "p"),
    "rpb_lift_bwd_rows": (_I, ""),
    "rpb_lift_bwd": (_I, "pppppp" + "iiiiiiiii" + "p"),
    "rpb_axis_gemm": (_I, "ppp" + "iiii" + "llll" + "ii" + "ppppi" + "p"),
    "rpb_mode_contract_fwd": (_I, "ppp" + "iii" + "p"),
    "rpb_mode_contract_dgrad": (_I, "ppp" + "iii" + "p"),
    "rpb_mode_contract_wgrad": (_I, "ppp" + "iiii" + "p"),
    "rpb_cell_mix_stat_rows": (_L, "liiiiii"),
    "rpb_cell_mix_wgrad_slots": (_L, "li"),
    "rpb_cell_mix_wgrad_supported": (_I, "lii"),
    "rpb_cell_mix_wgrad": (_I, "ppppppp" + "lii" + "ppppp" + "i" + "p"),
    "rpb_cell_mix_writes_gz": (_I, "liiiiii"),
    "rpb_cmx_debug_wave_times": (_I, "p"),
    "rpb_line_claim_set": (_I, "i"),
    "rpb_cell_mix": (_I, "ppppppp" + "l" + "iiii" + "ii" + "iiiiii" + "ppppi" + "pppppi" + "p"),
    "rpb_cell_wgrad_slots": (_L, "lii"),
    "rpb_cell_wgrad": (_I, "ppp" + "l" + "iii" + "iiiiii" + "ppppi" + "p"),
    "rpb_reduce_partials": (_I, "p" + "lll" + "pp" + "d" + "i" + "p"),
    "rpb_reduce_partials_batched": (_I, "p" + "i" + "llll" + "p" + "p"),
    "rpb_reduce_partials_grouped": (_I, "pilp"),
    "rpb_reduce_partials_grouped_cols": (_I, ""),
    "rpb_bn_finalize": (_I, "p" + "d" + "ff" + "pppp" + "i" + "p"),
    "rpb_bn_eval_prep": (_I, "p" + "f" + "p" + "i" + "p"),
    "rpb_bn_act_fwd": (_I, "pppppp" + "l" + "ii" + "p"),
    "rpb_bn_bwd_rows": (_I, ""),
    "rpb_bn_bwd_reduce": (_I, "ppppppp" + "l" + "ii" + "p"),
    "rpb_bn_bwd_apply": (_I, "ppppppp" + "d" + "p" + "l" + "ii" + "p"),
    "rpb_bn_bwd_row_slots": (_L, "i"),
    "rpb_bn_bwd_row_c128_supported": (_I, "ii"),
    "rpb_bn_bwd_row_c128": (_I, "pppppppp" + "d" + "i" + "ppp" + "iii" + "p"),
    "rpb_bn_bwd_row": (_I, "ppppppppp" + "d" + "i" + "ppppi" + "ppp" + "iiii" + "p"),
    "rpb_proj_slots": (_L, "lii"),
    "rpb_proj_fwd": (_I, "pppppp" + "l" + "ii" + "iiiiii" + "ppppi" + "i" + "p"),
    "rpb_proj_bwd": (_I, "pppppppp" + "l" + "ii" + "iiiiii" + "ppppi" + "i" + "p"),
    "rpb_mse_rows": (_I, ""),
    "rpb_mse": (_I, "ppppp" + "l" + "f" + "p"),
    "rpb_adam_step": (_I, "pppp" + "l" + "ffff" + "l" + "f" + "p"),
    "rpb_adam_step_ranges": (_I, "pppp" + "pil" + "ffff" + "l" + "f" + "p"),
    "rpb_rollout_affine": (_I, "ppp" + "l" + "ii" + "pppp" + "p"),
    "rpb_channel_affine": (_I, "pp" + "l" + "i" + "pp" + "i" + "p"),
    "rpb_gemm_nt": (_I, "pppppp" + "l" + "iiiii" + "ppp" + "iiiii" + "lf" + "p"),
    "rpb_mul": (_I, "ppp" + "l" + "p"),
    "rpb_add": (_I, "ppp" + "l" + "p"),
    "rpb_copy_cols": (_I, "pp" + "l" + "iiiii" + "p"),
    "rpb_gemm3x_tn_supported": (_I, "liiii"),
    "rpb_gemm3x_tn_splits": (_I, "lii"),
    "rpb_gemm3x_tn": (_I, "ppp" + "l" + "iiii" + "p"),
    "rpb_gemm_tn_splits": (_I, "liii"),
    "rpb_gemm_tn": (_I, "ppp" + "l" + "iiii" + "iiii" + "p"),
    "rpb_layernorm_bwd_rows": (_L, "l"),
    "rpb_layernorm_bwd": (_I, "pppppp" + "l" + "i" + "f" + "p"),
    "rpb_slice_bwd": (_I, "pppppppppp" + "iiii" + "p"),
    "rpb_colsum_rows": (_I, ""),
    "rpb_colsum": (_I, "pp" + "l" + "ii" + "p"),
    "rpb_tokens_lift": (_I, "pppp" + "l" + "iii" + "p"),
    "rpb_layernorm_fwd": (_I, "pppp" + "l" + "i" + "f" + "p"),
    "rpb_slice_blocks_per_sample": (_I, "i"),
    "rpb_slice_fwd": (_I, "ppppppp" + "iiiii" + "p" + "p"),
    "rpb_slice_attn": (_I, "pppppp" + "ii" + "p"),
    "rpb_slice_attn_train": (_I, "ppppppppppp" + "ii" + "p"),
    "rpb_deslice_fwd": (_I, "ppp" + "iiii" + "p"),
    "rpb_headnorm_fwd": (_I, "pipppi" + "l" + "i" + "f" + "p"),
    "rpb_headnorm_bwd_rows": (_L, "l"),
    "rpb_headnorm_bwd": (_I, "pippipip" + "l" + "i" + "f" + "p"),
    "rpb_head_scores_chunks": (_I, "il"),
    "rpb_head_scores": (_I, "pipip" + "ili" + "p"),
    "rpb_head_apply": (_I, "pippipipi" + "ili" + "lf" + "p"),
    "rpb_dropout_mul": (_I, "pp" + "l" + "lf" + "p"),
    "rpb_split3": (_I, "pp" + "lii" + "p"),
    "rpb_conv3x_wprep": (_I, "pp" + "ii" + "p"),
    "rpb_conv3x": (_I, "pppp" + "liii" + "iii" + "p"),
    "rpb_split3t": (_I, "pp" + "lii" + "iiii" + "p"),
    "rpb_gemm3x_wprep": (_I, "pp" + "ii" + "p"),
    "rpb_gemm3x": (_I, "pppppp" + "liiii" + "i" + "ppp" + "lf" + "p"),
    "rpb_conv3x_wgrad_splits": (_I, "lii"),
    "rpb_conv3x_wgrad": (_I, "ppp" + "lii" + "iii" + "p"),
    "rpb_window_pack": (_I, "ppppp" + "iiiiiiiiii" + "pppp" + "p"),
    "rpb_pair_pack": (_I, "ppppp" + "ilii" + "pppp" + "p"),
    "rpb_chan_blocks": (_I, "il"),
    "rpb_chan_stats": (_I, "pp" + "ili" + "p"),
    "rpb_affine_silu_fwd": (_I, "ppppp" + "ili" + "p"),
    "rpb_affine_silu_bwd_reduce": (_I, "ppppp" + "ili" + "p"),
    "rpb_affine_silu_bwd_apply": (_I, "ppppppp" + "ili" + "p"),
    "rpb_gn_affine_fwd": (_I, "pppp" + "df" + "ppp" + "iii" + "p"),
    "rpb_gn_affine_bwd": (_I, "ppppp" + "d" + "ppppp" + "iii" + "p"),
    "rpb_silu_fwd": (_I, "pp" + "l" + "p"),
    "rpb_silu_bwd": (_I, "ppp" + "l" + "p"),
    "rpb_relpos_bias_fwd": (_I, "ppp" + "ii" + "p"),
    "rpb_relpos_bias_bwd": (_I, "ppp" + "iii" + "p"),
    "rpb_im2col": (_I, "pp" + "iiiiiii" + "p"),
    "rpb_tattn_blocks": (_I, "l"),
    "rpb_tattn_fwd": (_I, "ppppp" + "iii" + "p"),
    "rpb_tattn_bwd": (_I, "ppppppp" + "iii" + "p"),
    "rpb_sattn_fwd": (_I, "ppp" + "ii" + "p"),
    "rpb_sattn_bwd": (_I, "ppppp" + "ii" + "p"),
    "rpb_linattn_prep_fwd": (_I, "ppp" + "ii" + "p"),
    "rpb_linattn_prep_bwd": (_I, "pppp" + "ii" + "p"),
    "rpb_col_reduce": (_I, "pip" + "ilii" + "p"),
    "rpb_pad_grid_fwd": (_I, "ppppppp" + "iiiiiiii" + "p"),
    "rpb_crop_gather": (_I, "pp" + "iiiiiiii" + "p"),
}

_lib = None


class RpbError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (once).  Raises if it has not been built -- no silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RpbError(
                f"{LIB_PATH} is missing: build it with `python -m realpdebench_amd.build` "
                "(hipcc --offload-arch=gfx950).  realpdebench_amd has no CPU/eager fallback.")
        # torch bundles its own libamdhip64: import it first so that this library binds to the SAME HIP runtime
        # instance (two runtimes in one process cannot share streams / device pointers).
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        lib.rpb_abi_version.restype = ctypes.c_int
        if lib.rpb_abi_version() != ABI_VERSION:            # the signature table below is for ONE version of include/rpb.h
            raise RpbError(f"{LIB_PATH} reports ABI version {lib.rpb_abi_version()}, this binding is for {ABI_VERSION}: rebuild the library")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = [_T[c] for c in args]
        _lib = lib
    return _lib


# ---- optional per-launch HIP-event timing (bench.py): PROFILE = {label: [(start, end, bytes, flops), ...]}
PROFILE = None
PROFILE_ONLY = None      # restrict timing to these labels (None = all)


def call(name, *args, label=None, nbytes=0, flops=0):
    """Call an ``int``-returning entry point and turn a non-zero status into ``RpbError``.

    ``label`` / ``nbytes`` / ``flops`` describe the launch for the roofline bookkeeping: algorithmic HBM bytes and
    floating-point operations of this launch."""
    lib = load()
    prof = PROFILE is not None and (PROFILE_ONLY is None or (label or name) in PROFILE_ONLY)
    if prof:
        import torch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    rc = getattr(lib, name)(*args)
    if prof:
        e.record()
        PROFILE.setdefault(label or name, []).append((s, e, nbytes, flops))
    if rc != 0:
        raise RpbError(f"{name} failed ({rc}): {lib.rpb_last_error().decode()}")


def profile_summary():
    """{label: dict(calls, total_ms, avg_ms, bytes, flops)} -- call after torch.cuda.synchronize()."""
    out = {}
    for label, recs in (PROFILE or {}).items():
        ms = [s.elapsed_time(e) for s, e, _, _ in recs]
        out[label] = dict(calls=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                          bytes=sum(r[2] for r in recs) / len(recs), flops=sum(r[3] for r in recs) / len(recs))
    return out


def query(name, *args):
    v = getattr(load(), name)(*args)
    if v < 0:
        raise RpbError(f"{name}{args} unsupported")
    return int(v)
