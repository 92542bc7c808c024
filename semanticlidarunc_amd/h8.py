"""Checked wrappers of the fp16 channel-blocked ("h8") inference kernels (include/slu.h, section "h8").

An h8 activation is a contiguous ``torch.float16`` tensor of shape ``[N, G, H, W, 8]`` holding channels
``8g .. 8g+7`` of pixel (h, w) in its last axis (pad channels are zero).  All arithmetic happens in the HIP
kernels (fp16 operands, fp32 accumulate / epilogue); torch only owns the memory.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import NamedTuple, Optional, Sequence

import torch

from . import _lib, ops
from ._lib import ConvH8Desc, check
from .ops import _ptr, _req, _stream


class H8Source(NamedTuple):
    tensor: torch.Tensor                    # [nimg, G, H, W, 8] fp16
    scale: Optional[torch.Tensor] = None    # [N, 8 G] fp32 multiplier (folded Dropout2d) or None
    nbatch: int = 0                         # > 0: tensor holds nbatch images, output image n reads image n % nbatch


def _req_h8(t: torch.Tensor, name: str) -> torch.Tensor:
    _req(t, name, torch.float16)
    if t.dim() != 5 or t.shape[4] != 8:
        raise RuntimeError(f"{name}: expected an h8 tensor [N, G, H, W, 8], got {tuple(t.shape)}")
    return t


def to_h8(x: torch.Tensor, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 NCHW -> h8 (channels padded to a multiple of 8 with zeros), optionally times scale[n, c]."""
    _req(x, "x")
    if x.dim() != 4:
        raise RuntimeError(f"x: expected NCHW, got {tuple(x.shape)}")
    n, c, h, w = x.shape
    if scale is not None:
        _req(scale, "scale")
        if tuple(scale.shape) != (n, c):
            raise RuntimeError(f"scale: expected {(n, c)}, got {tuple(scale.shape)}")
    y = torch.empty((n, (c + 7) // 8, h, w, 8), dtype=torch.float16, device=x.device)
    check(_lib.load().slu_nchw_to_h8(x.data_ptr(), _ptr(scale), y.data_ptr(), n, c, h, w, _stream()), "slu_nchw_to_h8")
    return y


def from_h8(x: torch.Tensor, channels: Optional[int] = None) -> torch.Tensor:
    """h8 -> fp32 NCHW with `channels` channels (default 8 G)."""
    _req_h8(x, "x")
    n, g, h, w, _ = x.shape
    c = 8 * g if channels is None else int(channels)
    if not 8 * (g - 1) < c <= 8 * g:
        raise RuntimeError(f"from_h8: {c} channels do not fit {g} blocks")
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_h8_to_nchw(x.data_ptr(), y.data_ptr(), n, c, h, w, _stream()), "slu_h8_to_nchw")
    return y


def pack_conv_weight_h8(weight: torch.Tensor) -> torch.Tensor:
    """OIHW fp32 weight -> fp16 MFMA A-fragment image (re-run after every weight update)."""
    _req(weight, "weight")
    if weight.dim() != 4 or weight.shape[2] != weight.shape[3]:
        raise RuntimeError(f"weight: expected [Cout,Cin,k,k], got {tuple(weight.shape)}")
    lib = _lib.load()
    cout, cin, ks, _ = weight.shape
    nbytes = lib.slu_packed_weight_bytes_h8(cout, cin, ks)
    if nbytes == 0:
        raise RuntimeError("pack_conv_weight_h8: unsupported weight shape")
    out = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    check(lib.slu_pack_conv_weight_h8(weight.data_ptr(), cout, cin, ks, out.data_ptr(), _stream()), "slu_pack_conv_weight_h8")
    return out


def conv2d_h8(srcs: Sequence[H8Source], wpack: torch.Tensor, cin: int, cout: int, ksize: int, dil: int, pad: int,
              bias: Optional[torch.Tensor] = None, slope: Optional[float] = None,
              bn_a: Optional[torch.Tensor] = None, bn_b: Optional[torch.Tensor] = None,
              resid: Optional[torch.Tensor] = None, out_f32_nchw: bool = False) -> torch.Tensor:
    """out = [resid +] bn_a * leaky(conv(cat(srcs * scale)) + bias) + bn_b  (slu_conv2d_h8_fwd).
    `cin` = real input channels the weight was packed with (only the last source may carry pad channels)."""
    lib = _lib.load()
    if not 1 <= len(srcs) <= _lib.MAX_SRC:
        raise RuntimeError(f"conv2d_h8: 1..{_lib.MAX_SRC} sources supported, got {len(srcs)}")
    d = ConvH8Desc()
    n = h = w = None
    gin = 0
    keep = []
    for i, s in enumerate(srcs):
        t = _req_h8(s.tensor, f"src[{i}]")
        sn, sg, sh, sw, _ = t.shape
        if s.nbatch:
            if i == 0 or sn != s.nbatch or n % sn:
                raise RuntimeError(f"src[{i}]: a batch-broadcast source must follow a full-batch source and divide N")
            sn = n
        if n is None:
            n, h, w = sn, sh, sw
        elif (sn, sh, sw) != (n, h, w):
            raise RuntimeError(f"src[{i}]: spatial/batch size {(sn, sh, sw)} != {(n, h, w)}")
        if s.scale is not None:
            _req(s.scale, f"src[{i}].scale")
            if tuple(s.scale.shape) != (n, 8 * sg):
                raise RuntimeError(f"src[{i}].scale: expected {(n, 8 * sg)}, got {tuple(s.scale.shape)}")
        d.src[i].ptr, d.src[i].scale, d.src[i].G, d.src[i].nbatch = t.data_ptr(), _ptr(s.scale), sg, int(s.nbatch)
        gin += sg
        keep.append(t)
    if not 8 * (gin - 1) < cin <= 8 * gin:
        raise RuntimeError(f"conv2d_h8: cin={cin} does not match {gin} input blocks")
    _req(wpack, "wpack", torch.uint8)
    if wpack.numel() != lib.slu_packed_weight_bytes_h8(cout, cin, ksize):
        raise RuntimeError(f"wpack: {wpack.numel()} bytes does not match Cout={cout} Cin={cin} k={ksize} (h8)")
    for t, nme in ((bias, "bias"), (bn_a, "bn_a"), (bn_b, "bn_b")):
        if t is not None:
            _req(t, nme)
            if t.numel() != cout:
                raise RuntimeError(f"{nme}: expected {cout} elements, got {t.numel()}")
    if (bn_a is None) != (bn_b is None):
        raise RuntimeError("bn_a and bn_b must be given together")
    gout = (cout + 7) // 8
    if out_f32_nchw:
        if resid is not None:
            raise RuntimeError("conv2d_h8: no residual on the fp32 NCHW output form")
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=keep[0].device)
    else:
        out = torch.empty((n, gout, h, w, 8), dtype=torch.float16, device=keep[0].device)
    if resid is not None:
        _req_h8(resid, "resid")
        if tuple(resid.shape) != (n, gout, h, w, 8):
            raise RuntimeError(f"resid: expected {(n, gout, h, w, 8)}, got {tuple(resid.shape)}")
    d.nsrc = len(srcs)
    d.N, d.H, d.W, d.Cout = n, h, w, cout
    d.ksize, d.dil, d.pad = ksize, dil, pad
    d.wpack, d.bias = wpack.data_ptr(), _ptr(bias)
    d.has_act, d.slope = (0, 0.0) if slope is None else (1, float(slope))
    d.bn_a, d.bn_b, d.resid, d.out = _ptr(bn_a), _ptr(bn_b), _ptr(resid), out.data_ptr()
    d.out_f32_nchw = 1 if out_f32_nchw else 0
    if ops.TIMING is None:
        check(lib.slu_conv2d_h8_fwd(C.byref(d), _stream()), "slu_conv2d_h8_fwd")
        return out
    # measurement mode (bench.py): HIP events on the launch stream around this one kernel
    buf = C.create_string_buffer(96)
    check(lib.slu_conv2d_h8_kernel_name(C.byref(d), buf, 96), "slu_conv2d_h8_kernel_name")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.slu_conv2d_h8_fwd(C.byref(d), _stream()), "slu_conv2d_h8_fwd")
    e1.record()
    flops = 2.0 * cin * cout * ksize * ksize * n * h * w
    nbytes = n * h * w * (2.0 * cin + (4.0 if out_f32_nchw else 2.0) * cout) + 2.0 * cout * cin * ksize * ksize
    ops.TIMING.append((buf.value.decode(), flops, nbytes, e0, e1, nbytes + (n * h * w * 2.0 * cout if resid is not None else 0.0)))
    ops.TIMING_TAGS.append(f"N{n} {cin}->{cout} k{ksize}d{dil} {h}x{w}")
    return out


def conv_tail_supported(channels: int, h: int, w: int) -> bool:
    return bool(_lib.load().slu_conv_tail_h8_supported(int(channels), int(h), int(w)))


def conv_tail_shortcut_supported(channels: int, cin: int) -> bool:
    return bool(_lib.load().slu_conv_tail_h8_shortcut_supported(int(channels), int(cin)))


def conv_tail_h8(a1: torch.Tensor, a2: torch.Tensor, w2x2: torch.Tensor, w1x1: torch.Tensor,
                 bias_a: Optional[torch.Tensor], slope_a: Optional[float], bn_a: Optional[tuple],
                 bias_b: Optional[torch.Tensor], slope_b: Optional[float], bn_b: Optional[tuple],
                 resid: Optional[torch.Tensor] = None, shortcut: Optional[tuple] = None) -> torch.Tensor:
    """The fused tail of a SalsaNext block (slu_conv_tail_h8_fwd):
        a3  = bn_a(leaky(conv2x2_dil2(a2) + bias_a));   out = [resid +] bn_b(leaky(conv1x1(cat(a1, a2, a3)) + bias_b))
    a1 / a2 / resid: h8 [N, C/8, H, W, 8]; w2x2 / w1x1: pack_conv_weight_h8 of [C, C, 2, 2] / [C, 3C, 1, 1]; bn_*: (scale, shift) or None.
    shortcut = (x, wpack, bias, slope, cin) instead of resid: resid = leaky(conv1x1(x) + bias) computed inside the kernel from the block's input
    x (h8 [N, cin/8, H, W, 8], wpack = pack_conv_weight_h8 of [C, cin, 1, 1]); conv_tail_shortcut_supported(C, cin)."""
    lib = _lib.load()
    _req_h8(a1, "a1")
    _req_h8(a2, "a2")
    if a1.shape != a2.shape:
        raise RuntimeError(f"conv_tail_h8: a1 {tuple(a1.shape)} != a2 {tuple(a2.shape)}")
    n, g, h, w, _ = a1.shape
    c = 8 * g
    if not conv_tail_supported(c, h, w):
        raise RuntimeError(f"conv_tail_h8: C={c} is not covered by the fused kernel (use two conv2d_h8 calls)")
    _req(w2x2, "w2x2", torch.uint8)
    _req(w1x1, "w1x1", torch.uint8)
    if w2x2.numel() != lib.slu_packed_weight_bytes_h8(c, c, 2) or w1x1.numel() != lib.slu_packed_weight_bytes_h8(c, 3 * c, 1):
        raise RuntimeError("conv_tail_h8: packed weight sizes do not match C")
    d = _lib.ConvTailH8Desc()
    for name, t in (("bias_a", bias_a), ("bias_b", bias_b)) + tuple((f"bn_{k}[{i}]", v) for k, pair in (("a", bn_a), ("b", bn_b)) if pair is not None
                                                                       for i, v in enumerate(pair)):
        if t is not None:
            _req(t, name)
            if t.numel() != c:
                raise RuntimeError(f"{name}: expected {c} elements, got {t.numel()}")
    if resid is not None:
        _req_h8(resid, "resid")
        if resid.shape != a1.shape:
            raise RuntimeError(f"resid: expected {tuple(a1.shape)}, got {tuple(resid.shape)}")
    if shortcut is not None:
        sx, sw, sb, sslope, scin = shortcut
        if resid is not None:
            raise RuntimeError("conv_tail_h8: resid and shortcut are exclusive")
        if not conv_tail_shortcut_supported(c, scin):
            raise RuntimeError(f"conv_tail_h8: shortcut {scin} -> {c} is not covered (run the 1x1 conv and pass resid)")
        _req_h8(sx, "shortcut.x")
        _req(sw, "shortcut.w", torch.uint8)
        if tuple(sx.shape) != (n, scin // 8, h, w, 8) or sw.numel() != lib.slu_packed_weight_bytes_h8(c, scin, 1):
            raise RuntimeError("conv_tail_h8: shortcut operand sizes do not match")
        if sb is not None:
            _req(sb, "shortcut.bias")
            if sb.numel() != c:
                raise RuntimeError(f"shortcut.bias: expected {c} elements, got {sb.numel()}")
    out = torch.empty_like(a1)
    d.a1, d.a2, d.N, d.H, d.W, d.C = a1.data_ptr(), a2.data_ptr(), n, h, w, c
    if shortcut is not None:
        d.sc_x, d.sc_w, d.sc_bias, d.sc_cin = sx.data_ptr(), sw.data_ptr(), _ptr(sb), int(scin)
        d.sc_hasact, d.sc_slope = (0, 0.0) if sslope is None else (1, float(sslope))
    d.w2x2, d.w1x1 = w2x2.data_ptr(), w1x1.data_ptr()
    d.biasA, d.bnA_a, d.bnA_b = _ptr(bias_a), _ptr(None if bn_a is None else bn_a[0]), _ptr(None if bn_a is None else bn_a[1])
    d.biasB, d.bnB_a, d.bnB_b = _ptr(bias_b), _ptr(None if bn_b is None else bn_b[0]), _ptr(None if bn_b is None else bn_b[1])
    d.hasactA, d.slopeA = (0, 0.0) if slope_a is None else (1, float(slope_a))
    d.hasactB, d.slopeB = (0, 0.0) if slope_b is None else (1, float(slope_b))
    d.resid, d.out = _ptr(resid), out.data_ptr()
    if ops.TIMING is None:
        check(lib.slu_conv_tail_h8_fwd(C.byref(d), _stream()), "slu_conv_tail_h8_fwd")
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.slu_conv_tail_h8_fwd(C.byref(d), _stream()), "slu_conv_tail_h8_fwd")
    e1.record()
    # the layer-granular convention of SURVEY 8(d): both convs read their inputs and write their outputs once
    scin = shortcut[4] if shortcut is not None else 0
    flops = 2.0 * c * (c * (4 + 3) + scin) * n * h * w
    # layer-granular convention: every conv reads its inputs and writes its output once (the shortcut conv: x in, shortcut out)
    nbytes = n * h * w * 2.0 * (c + c + 3 * c + c + (scin + c if shortcut is not None else 0)) + 2.0 * c * (c * (4 + 3) + scin)
    # what the fused kernel must move: a1, a2 (+ the residual, or the shortcut's input) in, out once, weights
    min_bytes = n * h * w * 2.0 * (c * 3 + (c if resid is not None else 0) + scin) + 2.0 * c * (c * (4 + 3) + scin)
    res = 2 if shortcut is not None else (1 if resid is not None else 0)      # the instantiation slu_conv_tail_h8_fwd launches (rocprofv3 reports the same name)
    if os.environ.get("SLU_TAIL_V1") == "1" and shortcut is None:
        name = {32: "tail_h8_kernel<1, 1, 8, 2, true>", 64: "tail_h8_kernel<2, 1, 8, 1, true>", 128: "tail_h8_kernel<2, 2, 4, 1, false>"}[c]
    else:
        name = {32: f"tail2_h8_kernel<1, 2, 3, {res}>", 64: f"tail2_h8_kernel<2, 1, 4, {res}>", 128: "tail_h8_kernel<2, 2, 4, 1, false>"}[c]
    ops.TIMING.append((name, flops, nbytes, e0, e1, min_bytes))
    ops.TIMING_TAGS.append(f"N{n} {c}->{c} k2d2 + {3 * c}->{c} k1" + (f" + {scin}->{c} k1 shortcut" if shortcut is not None else "") + f" fused {h}x{w}")
    return out


def ctx_block_supported(cin: int, c: int, h: int, w: int) -> bool:
    return bool(_lib.load().slu_ctx_block_h8_supported(int(cin), int(c), int(h), int(w)))


def ctx_block_h8(x: torch.Tensor, cin: int, w1: torch.Tensor, w2: torch.Tensor, w3: torch.Tensor, bias1: Optional[torch.Tensor],
                 bias2: Optional[torch.Tensor], bn1: Optional[tuple], bias3: Optional[torch.Tensor], bn2: Optional[tuple],
                 slope: float) -> torch.Tensor:
    """One fused ResContextBlock (slu_ctx_block_h8_fwd, SalsaNext.py:25-39):
        s = leaky(conv1x1(x) + bias1);  a1 = bn1(leaky(conv3x3(s) + bias2));  out = s + bn2(leaky(conv3x3_dil2(a1) + bias3))
    x: h8 [N, ceil(cin/8), H, W, 8]; w1 / w2 / w3: pack_conv_weight_h8 of [32, cin, 1, 1] / [32, 32, 3, 3] / [32, 32, 3, 3]; bn*: (scale, shift)."""
    lib = _lib.load()
    _req_h8(x, "x")
    n, g, h, w, _ = x.shape
    if not 8 * (g - 1) < cin <= 8 * g or not ctx_block_supported(cin, 32, h, w):
        raise RuntimeError(f"ctx_block_h8: cin={cin} with {g} input blocks is not covered by the fused kernel")
    for t, nme, shape in ((w1, "w1", (32, cin, 1)), (w2, "w2", (32, 32, 3)), (w3, "w3", (32, 32, 3))):
        _req(t, nme, torch.uint8)
        if t.numel() != lib.slu_packed_weight_bytes_h8(*shape):
            raise RuntimeError(f"{nme}: packed weight size does not match {shape}")
    vecs = (("bias1", bias1), ("bias2", bias2), ("bias3", bias3)) + tuple((f"bn{k}[{i}]", v) for k, pair in ((1, bn1), (2, bn2)) if pair is not None
                                                                           for i, v in enumerate(pair))
    for name, t in vecs:
        if t is not None:
            _req(t, name)
            if t.numel() != 32:
                raise RuntimeError(f"{name}: expected 32 elements, got {t.numel()}")
    out = torch.empty((n, 4, h, w, 8), dtype=torch.float16, device=x.device)
    d = _lib.CtxBlockH8Desc()
    d.x, d.N, d.H, d.W, d.Cin, d.C = x.data_ptr(), n, h, w, int(cin), 32
    d.w1, d.w2, d.w3 = w1.data_ptr(), w2.data_ptr(), w3.data_ptr()
    d.bias1, d.bias2, d.bias3 = _ptr(bias1), _ptr(bias2), _ptr(bias3)
    d.bn1_a, d.bn1_b = _ptr(None if bn1 is None else bn1[0]), _ptr(None if bn1 is None else bn1[1])
    d.bn2_a, d.bn2_b = _ptr(None if bn2 is None else bn2[0]), _ptr(None if bn2 is None else bn2[1])
    d.slope, d.out = float(slope), out.data_ptr()
    if ops.TIMING is None:
        check(lib.slu_ctx_block_h8_fwd(C.byref(d), _stream()), "slu_ctx_block_h8_fwd")
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.slu_ctx_block_h8_fwd(C.byref(d), _stream()), "slu_ctx_block_h8_fwd")
    e1.record()
    px = float(n * h * w)
    flops = 2.0 * (cin * 32 + 2 * 9 * 32 * 32) * px
    # SURVEY 8(d) layer-granular bytes of the three convs (each reads its input and writes its output once) vs what the fused kernel moves
    nbytes = px * 2.0 * ((8 * g + 32) + (32 + 32) + (32 + 32)) + 2.0 * (32 * cin + 2 * 9 * 32 * 32)
    min_bytes = px * 2.0 * (8 * g + 32) + 2.0 * (32 * cin + 2 * 9 * 32 * 32)
    ops.TIMING.append((f"ctx_h8_kernel<{1 if cin <= 16 else 2}>", flops, nbytes, e0, e1, min_bytes))
    ops.TIMING_TAGS.append(f"N{n} {cin}->32 k1 + 32->32 k3d1 + 32->32 k3d2 fused {h}x{w}")
    return out


def head_mc_h8(x: torch.Tensor, wpack: torch.Tensor, bias: Optional[torch.Tensor], classes: int, passes: int, batch: int, eps: float = 1e-12):
    """x: h8 [T*B, G, H, W, 8] (pass-major) -> (p_bar [B,C,H,W], H_norm [B,H,W], MI_norm [B,H,W], preds int64 [B,H,W]): the 1x1 head
    conv and the MC-dropout reduction of trainer.py:1143-1154 in one launch (slu_head_mc_h8)."""
    _req_h8(x, "x")
    n, g, h, w, _ = x.shape
    if n != passes * batch:
        raise RuntimeError(f"head_mc_h8: {n} images != T * B = {passes} * {batch}")
    lib = _lib.load()
    _req(wpack, "wpack", torch.uint8)
    if wpack.numel() != lib.slu_packed_weight_bytes_h8(classes, 8 * g, 1):
        raise RuntimeError("head_mc_h8: packed weight size does not match the head")
    if bias is not None:
        _req(bias, "bias")
        if bias.numel() != classes:
            raise RuntimeError(f"bias: expected {classes} elements, got {bias.numel()}")
    dev = x.device
    p_bar = torch.empty((batch, classes, h, w), dtype=torch.float32, device=dev)
    hn = torch.empty((batch, h, w), dtype=torch.float32, device=dev)
    mi = torch.empty((batch, h, w), dtype=torch.float32, device=dev)
    preds = torch.empty((batch, h, w), dtype=torch.int64, device=dev)
    args = (x.data_ptr(), passes, batch, g, h * w, wpack.data_ptr(), _ptr(bias), classes, float(eps), p_bar.data_ptr(), hn.data_ptr(),
            mi.data_ptr(), preds.data_ptr(), _stream())
    if ops.TIMING is None:
        check(lib.slu_head_mc_h8(*args), "slu_head_mc_h8")
        return p_bar, hn, mi, preds
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.slu_head_mc_h8(*args), "slu_head_mc_h8")
    e1.record()
    # layer-granular accounting: the head conv reads its fp16 input and "writes" fp32 logits; the epilogue of SURVEY 8(d) reads them back
    flops = 2.0 * 8 * g * classes * n * h * w
    nbytes = n * h * w * (2.0 * 8 * g + 4.0 * classes) + 2.0 * classes * 8 * g
    # what the fused kernel must move: the decoder output of every pass in, p_bar / H / MI / argmax of every scan out
    min_bytes = n * h * w * 2.0 * 8 * g + batch * h * w * (4.0 * classes + 16.0) + 2.0 * classes * 8 * g
    ops.TIMING.append((f"head_mc_h8_kernel<{g // 2}>", flops, nbytes, e0, e1, min_bytes))
    ops.TIMING_TAGS.append(f"N{n} {8 * g}->{classes} k1 head + MC reduce T={passes} {h}x{w}")
    return p_bar, hn, mi, preds


def avgpool3s2_h8(x: torch.Tensor, scale: Optional[torch.Tensor] = None, n_out: Optional[int] = None) -> torch.Tensor:
    """AvgPool2d(3, 2, 1) of x[n % B] * scale[n] for n < n_out (n_out = B unless x is shared by stacked MC passes)."""
    _req_h8(x, "x")
    b, g, h, w, _ = x.shape
    n = b if n_out is None else int(n_out)
    if n % b:
        raise RuntimeError("avgpool3s2_h8: n_out must be a multiple of the input batch")
    if scale is not None:
        _req(scale, "scale")
        if tuple(scale.shape) != (n, 8 * g):
            raise RuntimeError(f"scale: expected {(n, 8 * g)}, got {tuple(scale.shape)}")
    y = torch.empty((n, g, (h + 1) // 2, (w + 1) // 2, 8), dtype=torch.float16, device=x.device)
    check(_lib.load().slu_avgpool3s2_h8(x.data_ptr(), _ptr(scale), y.data_ptr(), n, 0 if n == b else b, g, h, w, _stream()),
          "slu_avgpool3s2_h8")
    return y


def pixel_shuffle_h8(x: torch.Tensor, scale_in: Optional[torch.Tensor] = None, scale_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.PixelShuffle(2) of x * scale_in[n, c_in], times scale_out[n, c_out] (both Dropout2d multipliers around it)."""
    _req_h8(x, "x")
    n, g, h, w, _ = x.shape
    for t, nme, c in ((scale_in, "scale_in", 8 * g), (scale_out, "scale_out", 2 * g)):
        if t is not None:
            _req(t, nme)
            if tuple(t.shape) != (n, c):
                raise RuntimeError(f"{nme}: expected {(n, c)}, got {tuple(t.shape)}")
    y = torch.empty((n, (2 * g + 7) // 8, 2 * h, 2 * w, 8), dtype=torch.float16, device=x.device)
    check(_lib.load().slu_pixel_shuffle_h8(x.data_ptr(), _ptr(scale_in), _ptr(scale_out), y.data_ptr(), n, g, h, w, _stream()),
          "slu_pixel_shuffle_h8")
    return y
