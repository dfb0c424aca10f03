"""Tensor-level wrappers over the C ABI: torch provides device memory and the stream, the
HIP library does the arithmetic.  Used by the kernel parity tests and by the engine (which
pre-binds the same entry points into replayable plans)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import (EPI_BF16, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_GATE_RES, EPI_GELU, EPI_SILU, F32EPI_GATE_RES,  # noqa: F401
                   F32EPI_GELU, F32EPI_NONE, F32EPI_SILU, GemmF32Args, GemmNTArgs, GemmTNArgs, call)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def p(t):
    return None if t is None else t.data_ptr()


def _need(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f'{name}: expected contiguous CUDA {dtype}, got {t.dtype} cuda={t.is_cuda} contig={t.is_contiguous()}')


def gemm_nt(A, Bw, bias=None, epi=EPI_BF16, out=None, out2=None, outf=None, res=None, gate=None, gate_ld=0,
            rows_per_sample=1, aux=None, M=None, k_splits=0, colsum=None, no_out=False):
    """C = A[M,K] @ Bw[N,K]^T with a fused epilogue (see include/maskdit_hip.h).  `no_out`: leave the OPTIONAL bf16
    output of MDT_EPI_GELU / SILU / GATE_RES NULL (what the inference plans do)."""
    _need(A, torch.bfloat16, 'A')
    _need(Bw, torch.bfloat16, 'B')
    M = A.shape[0] if M is None else M
    N, K = Bw.shape[0], A.shape[1]
    a = GemmNTArgs()
    a.A, a.lda, a.B, a.ldb, a.M, a.N, a.K = p(A), A.stride(0), p(Bw), Bw.stride(0), M, N, K
    a.bias, a.epi = p(bias), epi
    if epi in (EPI_BF16, EPI_GELU, EPI_SILU, EPI_GATE_RES, EPI_DGELU, EPI_DSILU) and out is None and not (
            no_out and epi in (EPI_GELU, EPI_SILU, EPI_GATE_RES)):
        out = torch.empty(A.shape[0], N, device=A.device, dtype=torch.bfloat16)
    if epi in (EPI_GELU, EPI_SILU) and out2 is None:
        out2 = torch.empty(A.shape[0], N, device=A.device, dtype=torch.bfloat16)
    if epi in (EPI_F32, EPI_GATE_RES) and outf is None:
        outf = torch.empty(A.shape[0], N, device=A.device, dtype=torch.float32)
    a.out, a.ldo = p(out), (out.stride(0) if out is not None else 0)
    a.out2, a.ldo2 = p(out2), (out2.stride(0) if out2 is not None else 0)
    a.outf, a.ldof = p(outf), (outf.stride(0) if outf is not None else 0)
    a.res, a.ldres = p(res), (res.stride(0) if res is not None else 0)
    a.gate, a.gate_ld, a.rows_per_sample = p(gate), gate_ld, rows_per_sample
    a.aux, a.ldaux = p(aux), (aux.stride(0) if aux is not None else 0)
    a.k_splits = k_splits
    a.colsum = p(colsum)
    call('mdt_gemm_nt', C.byref(a), stream_ptr())
    return out, out2, outf


def gemm_tn(A, Bm, Cout, n1_valid=0, n2_valid=0, splits=0, N1=None, N2=None, colsum_a=None):
    """Cout[N1,N2] += A[M,N1]^T @ Bm[M,N2] (f32 atomics); colsum_a[N1] += column sums of A (optional)."""
    _need(A, torch.bfloat16, 'A')
    _need(Bm, torch.bfloat16, 'B')
    a = GemmTNArgs()
    a.A, a.lda, a.B, a.ldb = p(A), A.stride(0), p(Bm), Bm.stride(0)
    a.M, a.N1, a.N2 = A.shape[0], (N1 or A.shape[1]), (N2 or Bm.shape[1])
    a.C, a.ldc, a.n1_valid, a.n2_valid, a.splits = p(Cout), Cout.stride(0), n1_valid, n2_valid, splits
    a.colsum_a = p(colsum_a) if colsum_a is not None else None
    call('mdt_gemm_tn', C.byref(a), stream_ptr())
    return Cout


def attn_fwd(qkv, B, L, H, hd, L_valid=0):
    out = torch.empty(B * L, H * hd, device=qkv.device, dtype=torch.bfloat16)
    lse = torch.empty(B * H * L, device=qkv.device, dtype=torch.float32)
    call('mdt_attn_fwd', p(qkv), p(out), p(lse), B, L, H, hd, L_valid, stream_ptr())
    return out, lse


def attn_bwd(qkv, out, dout, lse, B, L, H, hd, L_valid=0):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    call('mdt_attn_bwd', p(qkv), p(out), p(dout), p(lse), p(delta), p(dqkv), B, L, H, hd, L_valid, stream_ptr())
    return dqkv


def ln_modulate_fwd(x, shift, scale, mod_ld, rows_per_sample):
    M, D = x.shape
    xn = torch.empty(M, D, device=x.device, dtype=torch.bfloat16)
    stats = torch.empty(M, 2, device=x.device, dtype=torch.float32)
    call('mdt_ln_modulate_fwd', p(x), p(shift), p(scale), mod_ld, rows_per_sample, p(xn), p(stats), M, D, stream_ptr())
    return xn, stats


def ln_modulate_fwd_res(xres, y, gate, gate_ld, shift, scale, mod_ld, rows_per_sample):
    """x = xres + gate[b] * y; (xn, stats) = ln_modulate_fwd(x) in one pass (mdt_ln_modulate_fwd_res)."""
    M, D = xres.shape
    x = torch.empty(M, D, device=xres.device, dtype=torch.float32)
    xn = torch.empty(M, D, device=xres.device, dtype=torch.bfloat16)
    stats = torch.empty(M, 2, device=xres.device, dtype=torch.float32)
    call('mdt_ln_modulate_fwd_res', p(xres), p(y), p(gate), gate_ld, p(shift), p(scale), mod_ld, rows_per_sample, p(x), p(xn), p(stats),
         M, D, stream_ptr())
    return x, xn, stats


def ln_modulate_bwd(dxn, x, stats, scale, mod_ld, rows_per_sample, dx, accumulate, dshift, dscale, dmod_ld):
    M, D = x.shape
    call('mdt_ln_modulate_bwd', p(dxn), p(x), p(stats), p(scale), mod_ld, rows_per_sample, p(dx), int(accumulate),
         p(dshift), p(dscale), dmod_ld, M, D, stream_ptr())


def gate_bwd(dx, y, gate, mod_ld, rows_per_sample, dgate, dmod_ld, dbias=None):
    M, D = dx.shape
    dys = torch.empty(M, D, device=dx.device, dtype=torch.bfloat16)
    call('mdt_gate_bwd', p(dx), p(y), p(gate), mod_ld, rows_per_sample, p(dys), p(dgate), dmod_ld, p(dbias), M, D,
         stream_ptr())
    return dys


def colsum_bf16(x, out):
    call('mdt_colsum_bf16', p(x), x.stride(0), p(out), x.shape[0], x.shape[1], stream_ptr())
    return out


def mask_sort(noise, len_keep):
    B, T = noise.shape
    dev = noise.device
    ids_shuffle = torch.empty(B, T, device=dev, dtype=torch.int64)
    ids_restore = torch.empty(B, T, device=dev, dtype=torch.int64)
    mask = torch.empty(B, T, device=dev, dtype=torch.float32)
    ids32 = torch.empty(B, 2 * T, device=dev, dtype=torch.int32)
    call('mdt_mask_sort', p(noise), B, T, len_keep, p(ids_shuffle), p(ids_restore), p(mask), p(ids32), stream_ptr())
    return ids_shuffle, ids_restore, mask, ids32


# ---- fp32-faithful inference path (csrc/f32path.hip) ----------------------------------------------------------------
def gemm_f32(A, Bw, out, M, N, K, lda=None, ldb=None, ldo=None, bias=None, epi=F32EPI_NONE, res=None, gate=None, gate_ld=0,
             rows_per_sample=1, b_kmajor=False, batch=0, heads=0, a_strides=(0, 0), b_strides=(0, 0), o_strides=(0, 0),
             a_off=0, b_off=0, o_off=0):
    """out[z] = A[z][M,K] @ B[z]^T (+ bias) in exact fp32 (mdt_gemm_f32); *_off = element offsets into the tensors."""
    for t, nm in ((A, 'A'), (Bw, 'B'), (out, 'out')):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise ValueError(f'{nm}: expected a CUDA float32 tensor')
    a = GemmF32Args()
    a.A, a.lda = A.data_ptr() + 4 * a_off, (A.stride(0) if lda is None else lda)
    a.B, a.ldb, a.b_kmajor = Bw.data_ptr() + 4 * b_off, (Bw.stride(0) if ldb is None else ldb), int(b_kmajor)
    a.M, a.N, a.K = M, N, K
    a.bias, a.epi = p(bias), epi
    a.out, a.ldo = out.data_ptr() + 4 * o_off, (out.stride(0) if ldo is None else ldo)
    a.res, a.ldres = p(res), (res.stride(0) if res is not None else 0)
    a.gate, a.gate_ld, a.rows_per_sample = p(gate), gate_ld, rows_per_sample
    a.batch, a.heads = batch, heads
    a.a_stride_b, a.a_stride_h = a_strides
    a.b_stride_b, a.b_stride_h = b_strides
    a.o_stride_b, a.o_stride_h = o_strides
    call('mdt_gemm_f32', C.byref(a), stream_ptr())
    return out


def attention_f32(qkv, B, L, H, hd, three_launch=False):
    """timm Attention's softmax(q k^T hd^-0.5) v on a packed fp32 [B*L, 3*H*hd] buffer, as the fp32 plans run it
    (mdt_attn_f32: fused kernel where the shape allows).  three_launch=True: the general q k^T -> softmax -> p v form
    spelled out with the batched GEMM entry (what mdt_attn_f32 itself falls back to)."""
    W = H * hd
    o = torch.empty(B * L, W, device=qkv.device, dtype=torch.float32)
    if not three_launch:
        n = int(_lib.lib().mdt_attn_f32_ws_floats(B, L, H, hd))
        ws = torch.empty(n, device=qkv.device, dtype=torch.float32) if n else None
        call('mdt_attn_f32', p(qkv), p(o), p(ws), B, L, H, hd, stream_ptr())
        return o
    S = torch.empty(B * H * L, L, device=qkv.device, dtype=torch.float32)
    gemm_f32(qkv, qkv, S, L, L, hd, lda=3 * W, ldb=3 * W, ldo=L, batch=B * H, heads=H, a_strides=(L * 3 * W, hd),
             b_strides=(L * 3 * W, hd), o_strides=(H * L * L, L * L), b_off=W)
    call('mdt_softmax_rows_f32', p(S), B * H * L, L, L, float(hd) ** -0.5, stream_ptr())
    gemm_f32(S, qkv, o, L, hd, L, lda=L, ldb=3 * W, ldo=W, b_kmajor=True, batch=B * H, heads=H, a_strides=(H * L * L, L * L),
             b_strides=(L * 3 * W, hd), o_strides=(L * W, hd), b_off=2 * W)
    return o


def ln_modulate_f32(x, shift, scale, mod_ld, rows_per_sample):
    M, D = x.shape
    xn = torch.empty(M, D, device=x.device, dtype=torch.float32)
    call('mdt_ln_modulate_f32', p(x), p(shift), p(scale), mod_ld, rows_per_sample, p(xn), M, D, stream_ptr())
    return xn
