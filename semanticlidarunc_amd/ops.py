"""Thin, checked Python wrappers over the C ABI (include/slu.h).

torch tensors own the memory; every function validates device / dtype / contiguity / shape on the
host before the launch (a wrong shape must never reach a hand-written kernel) and launches on
torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional, Sequence

import torch

from . import _lib
from ._lib import ConvDesc, check

INT64_MIN = -(2 ** 63)

# bench.py sets this to a list to collect (kernel name, algorithmic flops, algorithmic bytes, ev0, ev1)
# per conv launch; None (the default) adds nothing to the launch path.
TIMING = None
TIMING_TAGS = []


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on '{t.device}'; the HIP path only runs on a GPU (no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        # launches go to the CURRENT device's stream; a tensor living on another GPU would be dereferenced there
        raise RuntimeError(f"{name}: tensor is on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device(local_rank) (distributed.init_from_env does) or use `with torch.cuda.device(t.device)`")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous")
    return t


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
class ConvSource(NamedTuple):
    """One channel-concatenated input of a fused conv."""
    tensor: torch.Tensor                    # [N,C,H,W], or [N,C,H/2,W/2] when pixel_shuffle
    scale: Optional[torch.Tensor] = None    # [N,C] multiplier (folded Dropout2d) or None
    pixel_shuffle: bool = False
    nbatch: int = 0                         # > 0: tensor holds nbatch images, output image n reads image n % nbatch
    cuse: int = 0                           # > 0: only the first cuse channels of the tensor contribute


def conv_ck(ksize: int) -> int:
    return _lib.load().slu_conv_ck(ksize)


def pack_conv_weight(weight: torch.Tensor) -> torch.Tensor:
    """OIHW fp32 weight -> MFMA A-fragment image (re-run after every weight update)."""
    _req(weight, "weight")
    if weight.dim() != 4 or weight.shape[2] != weight.shape[3]:
        raise RuntimeError(f"weight: expected [Cout,Cin,k,k], got {tuple(weight.shape)}")
    lib = _lib.load()
    cout, cin, ks, _ = weight.shape
    ck = lib.slu_conv_ck(ks)
    n = lib.slu_packed_weight_floats(cout, cin, ks, ck)
    if n == 0:
        raise RuntimeError("pack_conv_weight: unsupported weight shape")
    out = torch.empty(n, dtype=torch.float32, device=weight.device)
    check(lib.slu_pack_conv_weight(weight.data_ptr(), cout, cin, ks, ck, out.data_ptr(), _stream()), "slu_pack_conv_weight")
    return out


class WeightPackPlan:
    """The forward AND data-gradient MFMA images of a list of conv weights, rebuilt by ONE launch (slu_pack_conv_weights_multi) into buffers
    that keep their addresses: `fwd[i]` == pack_conv_weight(w_i), `dgrad[i]` == pack_conv_weight(dgrad_weight(w_i)).  The plan records the
    weights' addresses; `matches(weights)` tells whether it still describes them (a parameter that was re-allocated needs a new plan)."""

    def __init__(self, weights: Sequence[torch.Tensor]):
        lib = _lib.load()
        self.ptrs = [w.data_ptr() for w in weights]
        self.fwd, self.dgrad = [], []
        jobs = (_lib.PackJob * (2 * len(weights)))()
        begin = 0
        for i, w in enumerate(weights):
            _req(w, f"weights[{i}]")
            if w.dim() != 4 or w.shape[2] != w.shape[3]:
                raise RuntimeError(f"weights[{i}]: expected [Cout,Cin,k,k], got {tuple(w.shape)}")
            cout, cin, ks, _ = w.shape
            ck = lib.slu_conv_ck(ks)
            for d, (co, ci), store in ((0, (cout, cin), self.fwd), (1, (cin, cout), self.dgrad)):
                n = lib.slu_packed_weight_floats(co, ci, ks, ck)
                if n == 0:
                    raise RuntimeError(f"weights[{i}]: unsupported shape {tuple(w.shape)}")
                out = torch.empty(n, dtype=torch.float32, device=w.device)
                store.append(out)
                j = jobs[2 * i + d]
                j.w, j.out, j.cout, j.cin, j.ksize, j.ck, j.dgrad, j.begin = w.data_ptr(), out.data_ptr(), cout, cin, ks, ck, d, begin
                begin += n
        self.total, self.njobs = begin, 2 * len(weights)
        import ctypes
        raw = bytes(ctypes.string_at(ctypes.addressof(jobs), ctypes.sizeof(jobs)))
        self.jobs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(weights[0].device)

    def matches(self, weights: Sequence[torch.Tensor]) -> bool:
        return len(weights) == len(self.ptrs) and all(w.data_ptr() == p for w, p in zip(weights, self.ptrs))

    def run(self) -> None:
        check(_lib.load().slu_pack_conv_weights_multi(self.jobs.data_ptr(), self.njobs, self.total, _stream()), "slu_pack_conv_weights_multi")


def pack_conv_weight_f16x3(weight: torch.Tensor) -> torch.Tensor:
    """OIHW fp32 weight -> split-fp16 (hi, lo) MFMA A-fragment image for precision='f16x3'."""
    _req(weight, "weight")
    if weight.dim() != 4 or weight.shape[2] != weight.shape[3]:
        raise RuntimeError(f"weight: expected [Cout,Cin,k,k], got {tuple(weight.shape)}")
    lib = _lib.load()
    cout, cin, ks, _ = weight.shape
    n = lib.slu_packed_weight_bytes_f16x3(cout, cin, ks)
    if n == 0:
        raise RuntimeError("pack_conv_weight_f16x3: unsupported weight shape")
    out = torch.empty(n, dtype=torch.uint8, device=weight.device)
    check(lib.slu_pack_conv_weight_f16x3(weight.data_ptr(), cout, cin, ks, out.data_ptr(), _stream()), "slu_pack_conv_weight_f16x3")
    return out


PRECISIONS = {"fp32": 0, "f16x3": 1}


def bn_fold(gamma, beta, mean, var, eps: float):
    """(a, b) with  a*x + b == eval BatchNorm(x)."""
    c = gamma.numel()
    for t, nme in ((gamma, "gamma"), (beta, "beta"), (mean, "running_mean"), (var, "running_var")):
        _req(t, nme)
        if t.numel() != c:
            raise RuntimeError("bn_fold: size mismatch")
    a = torch.empty(c, dtype=torch.float32, device=gamma.device)
    b = torch.empty_like(a)
    check(_lib.load().slu_bn_fold(gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(), float(eps), c,
                                  a.data_ptr(), b.data_ptr(), _stream()), "slu_bn_fold")
    return a, b


def conv2d_fused(srcs: Sequence[ConvSource], wpack: torch.Tensor, cout: int, ksize: int, dil: int, pad: int,
                 bias: Optional[torch.Tensor] = None, slope: Optional[float] = None,
                 bn_a: Optional[torch.Tensor] = None, bn_b: Optional[torch.Tensor] = None,
                 resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 precision: str = "fp32", act: Optional[str] = None, act_after_resid: bool = False,
                 stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = resid + bn_a * leaky(conv(cat(srcs)) + bias) + bn_b   (see slu_conv2d_fwd).
    precision 'fp32' (exact, wpack from pack_conv_weight) or 'f16x3' (split-fp16, wpack from pack_conv_weight_f16x3)."""
    if precision not in PRECISIONS:
        raise ValueError(f"unknown conv precision {precision!r}")
    lib = _lib.load()
    if not 1 <= len(srcs) <= _lib.MAX_SRC:
        raise RuntimeError(f"conv2d_fused: 1..{_lib.MAX_SRC} sources supported, got {len(srcs)}")
    d = ConvDesc()
    n = h = w = None
    cin = 0
    keep = []
    for i, s in enumerate(srcs):
        t = _req(s.tensor, f"src[{i}]")
        if t.dim() != 4:
            raise RuntimeError(f"src[{i}]: expected NCHW, got {tuple(t.shape)}")
        sn, sc, sh, sw = t.shape
        if s.pixel_shuffle:
            if sc % 4:
                raise RuntimeError(f"src[{i}]: PixelShuffle(2) needs C % 4 == 0")
            sh, sw, contributed = sh * 2, sw * 2, sc // 4
        elif s.cuse:
            if not 0 < s.cuse <= sc:
                raise RuntimeError(f"src[{i}]: cuse must be in 1..C")
            contributed = s.cuse
        else:
            contributed = sc
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
            if tuple(s.scale.shape) != (sn, sc):
                raise RuntimeError(f"src[{i}].scale: expected {(sn, sc)}, got {tuple(s.scale.shape)}")
        d.src[i].nbatch = int(s.nbatch)
        d.src[i].cuse = int(s.cuse)
        d.src[i].ptr = t.data_ptr()
        d.src[i].scale = _ptr(s.scale)
        d.src[i].C = sc
        d.src[i].pixel_shuffle = 1 if s.pixel_shuffle else 0
        cin += contributed
        keep.append(t)
    if precision == "f16x3":
        ck = 16
        _req(wpack, "wpack", torch.uint8)
        if wpack.numel() != lib.slu_packed_weight_bytes_f16x3(cout, cin, ksize):
            raise RuntimeError(f"wpack: {wpack.numel()} bytes does not match Cout={cout} Cin={cin} k={ksize} (f16x3)")
    else:
        ck = lib.slu_conv_ck(ksize)
        _req(wpack, "wpack")
        if wpack.numel() != lib.slu_packed_weight_floats(cout, cin, ksize, ck):
            raise RuntimeError(f"wpack: {wpack.numel()} floats does not match Cout={cout} Cin={cin} k={ksize}")
    for t, nme in ((bias, "bias"), (bn_a, "bn_a"), (bn_b, "bn_b")):
        if t is not None:
            _req(t, nme)
            if t.numel() != cout:
                raise RuntimeError(f"{nme}: expected {cout} elements, got {t.numel()}")
    if (bn_a is None) != (bn_b is None):
        raise RuntimeError("bn_a and bn_b must be given together")
    if out is None:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=keep[0].device)
    else:
        _req(out, "out")
        if tuple(out.shape) != (n, cout, h, w):
            raise RuntimeError(f"out: expected {(n, cout, h, w)}, got {tuple(out.shape)}")
    if resid is not None:
        _req(resid, "resid")
        if tuple(resid.shape) != (n, cout, h, w):
            raise RuntimeError(f"resid: expected {(n, cout, h, w)}, got {tuple(resid.shape)}")
    d.nsrc = len(srcs)
    d.N, d.H, d.W, d.Cin, d.Cout = n, h, w, cin, cout
    d.ksize, d.dil, d.pad, d.ck = ksize, dil, pad, ck
    d.wpack, d.bias = wpack.data_ptr(), _ptr(bias)
    d.has_act, d.slope = (0, 0.0) if slope is None else (1, float(slope))
    if act is not None:                     # explicit activation kind overrides `slope`
        if act not in ("relu", "tanh", "silu", "none"):
            raise ValueError(f"unknown activation {act!r}")
        d.has_act, d.slope = {"relu": (1, 0.0), "tanh": (2, 0.0), "silu": (3, 0.0), "none": (0, 0.0)}[act]
        if act_after_resid and act in ("tanh", "silu"):
            raise ValueError(f"activation {act!r} has no after-the-residual form")
    if act_after_resid and d.has_act:
        d.has_act |= 4
    d.bn_a, d.bn_b, d.resid, d.out = _ptr(bn_a), _ptr(bn_b), _ptr(resid), out.data_ptr()
    d.precision = PRECISIONS[precision]
    if stats is not None:
        # per-channel sum / sum of squares of the stored output, added into `stats` (f64 [2, Cout], zeroed by the caller)
        _req(stats, "stats", torch.float64)
        if tuple(stats.shape) != (2, cout) or precision != "fp32":
            raise RuntimeError("stats: expected a float64 [2, Cout] tensor with precision='fp32'")
        d.stats = stats.data_ptr()
    if TIMING is None:
        check(lib.slu_conv2d_fwd(C.byref(d), _stream()), "slu_conv2d_fwd")
        return out
    # measurement mode (bench.py): HIP events on the launch stream around this one kernel
    buf = C.create_string_buffer(64)
    check(lib.slu_conv2d_kernel_name(C.byref(d), buf, 64), "slu_conv2d_kernel_name")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.slu_conv2d_fwd(C.byref(d), _stream()), "slu_conv2d_fwd")
    e1.record()
    flops = 2.0 * cin * cout * ksize * ksize * n * h * w
    nbytes = 4.0 * (n * h * w * (cin + cout) + cout * cin * ksize * ksize)
    TIMING.append((buf.value.decode(), flops, nbytes, e0, e1))
    TIMING_TAGS.append(f"N{n} {cin}->{cout} k{ksize}d{dil} {h}x{w}")
    return out


def avgpool3s2_bcast(x: torch.Tensor, scale: Optional[torch.Tensor], n_out: int) -> torch.Tensor:
    """avgpool3s2 of x[n % B] * scale[n] for n < n_out: pools a B-image tensor into n_out = T*B stacked passes."""
    _req(x, "x")
    b, c, h, w = x.shape
    if n_out % b:
        raise RuntimeError("avgpool3s2_bcast: n_out must be a multiple of the input batch")
    if scale is not None:
        _req(scale, "scale")
        if tuple(scale.shape) != (n_out, c):
            raise RuntimeError(f"scale: expected {(n_out, c)}, got {tuple(scale.shape)}")
    y = torch.empty((n_out, c, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_avgpool3s2_bcast_fwd(x.data_ptr(), _ptr(scale), y.data_ptr(), n_out, b, c, h, w, _stream()),
          "slu_avgpool3s2_bcast_fwd")
    return y


def avgpool3s2(x: torch.Tensor, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x, "x")
    n, c, h, w = x.shape
    if scale is not None:
        _req(scale, "scale")
        if tuple(scale.shape) != (n, c):
            raise RuntimeError(f"scale: expected {(n, c)}, got {tuple(scale.shape)}")
    y = torch.empty((n, c, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_avgpool3s2_fwd(x.data_ptr(), _ptr(scale), y.data_ptr(), n, c, h, w, _stream()), "slu_avgpool3s2_fwd")
    return y


# ------------------------------------------------------------------------------------------------
# uncertainty / loss / metrics
# ------------------------------------------------------------------------------------------------
def mc_reduce(mc_logits: torch.Tensor, eps: float = 1e-12):
    """[T,B,C,H,W] logits -> (p_bar[B,C,H,W], H_norm[B,H,W], MI_norm[B,H,W], preds[B,H,W] int64)."""
    _req(mc_logits, "mc_logits")
    if mc_logits.dim() != 5:
        raise RuntimeError(f"mc_logits: expected [T,B,C,H,W], got {tuple(mc_logits.shape)}")
    t, b, c, h, w = mc_logits.shape
    dev = mc_logits.device
    p_bar = torch.empty((b, c, h, w), dtype=torch.float32, device=dev)
    hn = torch.empty((b, h, w), dtype=torch.float32, device=dev)
    mi = torch.empty((b, h, w), dtype=torch.float32, device=dev)
    preds = torch.empty((b, h, w), dtype=torch.int64, device=dev)
    check(_lib.load().slu_mc_reduce(mc_logits.data_ptr(), t, b, c, h * w, float(eps), p_bar.data_ptr(), hn.data_ptr(),
                                    mi.data_ptr(), preds.data_ptr(), _stream()), "slu_mc_reduce")
    return p_bar, hn, mi, preds


def softmax_entropy(logits: torch.Tensor, eps: float = 1e-8):
    """[B,C,H,W] logits -> (probs, H_norm[B,H,W], preds[B,H,W] int64)."""
    _req(logits, "logits")
    b, c, h, w = logits.shape
    dev = logits.device
    probs = torch.empty_like(logits)
    hn = torch.empty((b, h, w), dtype=torch.float32, device=dev)
    preds = torch.empty((b, h, w), dtype=torch.int64, device=dev)
    check(_lib.load().slu_softmax_entropy(logits.data_ptr(), b, c, h * w, float(eps), probs.data_ptr(), hn.data_ptr(),
                                          preds.data_ptr(), _stream()), "slu_softmax_entropy")
    return probs, hn, preds


def confusion_update(confmat: torch.Tensor, preds: torch.Tensor, targets: torch.Tensor) -> None:
    """confmat[C,C] int64 (rows = GT) += bincount(t*C+p) over in-range pairs, in place."""
    _req(confmat, "confmat", torch.int64)
    _req(preds, "preds", torch.int64)
    _req(targets, "targets", torch.int64)
    if preds.numel() != targets.numel():
        raise RuntimeError("preds/targets size mismatch")
    c = confmat.shape[0]
    if confmat.dim() != 2 or confmat.shape[1] != c:
        raise RuntimeError("confmat must be [C,C]")
    check(_lib.load().slu_confusion_update(preds.data_ptr(), targets.data_ptr(), preds.numel(), c, confmat.data_ptr(),
                                           _stream()), "slu_confusion_update")


def ece_update(probs, labels, count, sum_correct, sum_conf, ignore_index=None) -> None:
    """Accumulate top-label calibration bins in place (count int64[nb], sums float64[nb])."""
    _req(probs, "probs")
    _req(labels, "labels", torch.int64)
    _req(count, "count", torch.int64)
    _req(sum_correct, "sum_correct", torch.float64)
    _req(sum_conf, "sum_conf", torch.float64)
    b, c, h, w = probs.shape
    if labels.numel() != b * h * w:
        raise RuntimeError("labels size mismatch")
    nb = count.numel()
    if sum_correct.numel() != nb or sum_conf.numel() != nb:
        raise RuntimeError("bin accumulator size mismatch")
    ign = INT64_MIN if ignore_index is None else int(ignore_index)
    check(_lib.load().slu_ece_update(probs.data_ptr(), labels.data_ptr(), b, c, h * w, ign, nb, count.data_ptr(),
                                     sum_correct.data_ptr(), sum_conf.data_ptr(), _stream()), "slu_ece_update")


def softmax_nll(logits: torch.Tensor, labels: torch.Tensor, clamp: float = 1e-8, want_probs: bool = True):
    """(probs or None, nll_sum float64[1]) : sum over pixels of -ln max(softmax(logits)[label], clamp)."""
    _req(logits, "logits")
    _req(labels, "labels", torch.int64)
    b, c, h, w = logits.shape
    if labels.numel() != b * h * w:
        raise RuntimeError("labels size mismatch")
    probs = torch.empty_like(logits) if want_probs else None
    acc = torch.zeros(1, dtype=torch.float64, device=logits.device)
    check(_lib.load().slu_softmax_nll_fwd(logits.data_ptr(), labels.data_ptr(), b, c, h * w, float(clamp), _ptr(probs),
                                          acc.data_ptr(), _stream()), "slu_softmax_nll_fwd")
    return probs, acc


# ------------------------------------------------------------------------------------------------
# loss kernels
# ------------------------------------------------------------------------------------------------
NLL_LOGITS, NLL_PROBS_CLAMP, NLL_PROBS_EPS, NLL_LOG_PROBS = 0, 1, 2, 3


def lovasz_class_mask(classes, c: int) -> int:
    """The kernel's class set for the reference's `classes` argument (lovasz.py:66): 0 = 'present'."""
    if classes == "present":
        return 0
    ids = range(c) if classes == "all" else [int(k) for k in classes]
    mask = 0
    for k in ids:
        if not 0 <= k < c:
            raise IndexError(f"Lovasz: class {k} out of range for {c} classes")      # the reference indexes probas[:, k]
        mask |= 1 << k
    if mask == 0:
        raise ValueError("Lovasz: empty class list")
    return mask


def lovasz_fwd(probs: torch.Tensor, labels: torch.Tensor, ignore_index=None, want_grad: bool = True, classes="present"):
    """(loss f32[1], n_summed f32[1], grad_probs [B,C,H,W] or None) -- Lovasz-Softmax over the present classes, all classes or a list."""
    _req(probs, "probs")
    _req(labels, "labels", torch.int64)
    b, c, h, w = probs.shape
    if labels.numel() != b * h * w:
        raise RuntimeError("labels size mismatch")
    lib = _lib.load()
    nbytes = lib.slu_lovasz_workspace_bytes(b, c, h * w)
    if nbytes == 0:
        raise RuntimeError("lovasz: unsupported shape")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=probs.device)
    loss = torch.empty(1, dtype=torch.float32, device=probs.device)
    npres = torch.empty(1, dtype=torch.float32, device=probs.device)
    grad = torch.empty_like(probs) if want_grad else None
    ign = INT64_MIN if ignore_index is None else int(ignore_index)
    check(lib.slu_lovasz_fwd(probs.data_ptr(), labels.data_ptr(), b, c, h * w, ign, lovasz_class_mask(classes, c), ws.data_ptr(), nbytes, loss.data_ptr(),
                             npres.data_ptr(), _ptr(grad), _stream()), "slu_lovasz_fwd")
    return loss, npres, grad


def nll_fwd(x: torch.Tensor, labels: torch.Tensor, kind: int, param: float = 0.0, ignore_index=None):
    """(nll_sum f64[1], count i64[1]) over pixels whose label is in [0,C) and != ignore_index."""
    _req(x, "x")
    _req(labels, "labels", torch.int64)
    b, c, h, w = x.shape
    if labels.numel() != b * h * w:
        raise RuntimeError("labels size mismatch")
    acc = torch.zeros(1, dtype=torch.float64, device=x.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=x.device)
    ign = INT64_MIN if ignore_index is None else int(ignore_index)
    check(_lib.load().slu_nll_fwd(x.data_ptr(), labels.data_ptr(), b, c, h * w, int(kind), float(param), ign, acc.data_ptr(),
                                  cnt.data_ptr(), _stream()), "slu_nll_fwd")
    return acc, cnt


def nll_bwd(x, labels, kind: int, param: float, ignore_index, gscale: torch.Tensor):
    """grad_x = gscale[0] * d(sum nll)/dx ; gscale is a 1-element fp32 device tensor."""
    _req(x, "x")
    _req(labels, "labels", torch.int64)
    _req(gscale, "gscale")
    b, c, h, w = x.shape
    g = torch.empty_like(x)
    ign = INT64_MIN if ignore_index is None else int(ignore_index)
    check(_lib.load().slu_nll_bwd(x.data_ptr(), labels.data_ptr(), b, c, h * w, int(kind), float(param), ign, gscale.data_ptr(),
                                  g.data_ptr(), _stream()), "slu_nll_bwd")
    return g


def softmax_loss_bwd(probs, labels=None, dense=None, w_dense: float = 1.0, w_nll: float = 0.0, clamp: float = 0.0, gout=None):
    """grad_logits = softmax backward of gout * (w_dense * dense - [c==y, p_y>=clamp] * w_nll / p_y)."""
    _req(probs, "probs")
    b, c, h, w = probs.shape
    if labels is not None:
        _req(labels, "labels", torch.int64)
    if dense is not None:
        _req(dense, "dense")
        if dense.shape != probs.shape:
            raise RuntimeError("dense gradient shape mismatch")
    if gout is not None:
        _req(gout, "gout")
    g = torch.empty_like(probs)
    check(_lib.load().slu_softmax_loss_bwd(probs.data_ptr(), _ptr(labels), _ptr(dense), float(w_dense), float(w_nll), float(clamp),
                                           _ptr(gout), b, c, h * w, g.data_ptr(), _stream()), "slu_softmax_loss_bwd")
    return g


# ------------------------------------------------------------------------------------------------
# training-side kernels
# ------------------------------------------------------------------------------------------------
def _fill_srcs(arr, srcs):
    n = h = w = None
    for i, s in enumerate(srcs):
        t = _req(s.tensor, f"src[{i}]")
        sn, sc, sh, sw = t.shape
        if s.pixel_shuffle:
            sh, sw = sh * 2, sw * 2
        if n is None:
            n, h, w = sn, sh, sw
        elif (sn, sh, sw) != (n, h, w):
            raise RuntimeError(f"src[{i}]: size mismatch")
        if s.scale is not None:
            _req(s.scale, f"src[{i}].scale")
            if tuple(s.scale.shape) != (sn, sc):
                raise RuntimeError(f"src[{i}].scale: expected {(sn, sc)}")
        arr[i].ptr, arr[i].scale, arr[i].C, arr[i].pixel_shuffle = t.data_ptr(), _ptr(s.scale), sc, 1 if s.pixel_shuffle else 0
    return n, h, w


class _ZeroArena:
    """fp64 zeros by the slice.  A training step needs ~140 tiny zero-initialised accumulators (BatchNorm sums, bias gradients): one fill
    launch each was 0.6 ms of a 34 ms step.  A chunk is zeroed ONCE and handed out slice by slice -- every slice is used exactly once, as
    the accumulator a kernel adds into; an exhausted chunk is dropped (the views handed out keep it alive) and a new one zeroed.  While a
    stream is being captured into a HIP graph the plain torch.zeros is used (its fill is part of the graph and re-runs on replay)."""
    CHUNK = 1 << 16

    def __init__(self):
        self._chunks = {}          # (device, stream) -> [tensor, next free element]

    def take(self, shape, device) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        if n > self.CHUNK // 4 or torch.cuda.is_current_stream_capturing():
            return torch.zeros(shape, dtype=torch.float64, device=device)
        # one chunk per (device, stream): the fill is ordered before the consumers only on the stream it was issued on, and the caching
        # allocator may hand a dropped chunk's block out again on that stream while another stream's kernels still add into it (the dropout
        # side stream and the graph warm-up stream of this package are real second streams)
        dev = torch.device(device)
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ent = self._chunks.get(key)
        if ent is None or ent[1] + n > self.CHUNK:
            ent = self._chunks[key] = [torch.zeros(self.CHUNK, dtype=torch.float64, device=device), 0]
        out = ent[0][ent[1]:ent[1] + n].view(shape)
        ent[1] += n
        return out


_ZEROS64 = _ZeroArena()


class _ZeroArenaF32(_ZeroArena):
    """The same for the fp32 accumulators of the weight-gradient kernels (split-K partial sums added with atomics: the packed scratch of the
    3x3 / 2x2 families, the gradient itself for the 1x1 family): SalsaNext's ~50 of them total 7 M floats per step, one fill instead of 50.
    Slices start on 256-byte boundaries."""
    CHUNK = 1 << 23

    def take(self, n: int, device) -> torch.Tensor:
        n = int(n)
        if n > self.CHUNK // 2 or torch.cuda.is_current_stream_capturing():
            return torch.zeros(n, dtype=torch.float32, device=device)
        dev = torch.device(device)
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ent = self._chunks.get(key)
        if ent is None or ent[1] + n > self.CHUNK:
            ent = self._chunks[key] = [torch.zeros(self.CHUNK, dtype=torch.float32, device=device), 0]
        out = ent[0][ent[1]:ent[1] + n]
        ent[1] += (n + 63) // 64 * 64
        return out


_ZEROS32 = _ZeroArenaF32()


def zeros_f64(shape, device) -> torch.Tensor:
    """A zero-filled float64 accumulator that is written once (see _ZeroArena)."""
    return _ZEROS64.take(tuple(shape) if not isinstance(shape, int) else (shape,), device)


def bn_stats(y: torch.Tensor):
    """(sum f64[C], sumsq f64[C]) over (N,H,W)."""
    _req(y, "y")
    n, c, h, w = y.shape
    sq = zeros_f64((2, c), y.device)
    s, q = sq[0], sq[1]
    check(_lib.load().slu_bn_stats(y.data_ptr(), n, c, h * w, s.data_ptr(), q.data_ptr(), _stream()), "slu_bn_stats")
    return s, q


def bn_bwd_reduce(dz, y, mean, invstd):
    """(sum dz, sum dz*xhat) per channel, f64."""
    for t, nme in ((dz, "dz"), (y, "y"), (mean, "mean"), (invstd, "invstd")):
        _req(t, nme)
    if dz.shape != y.shape:
        raise RuntimeError("dz / y shape mismatch")
    n, c, h, w = y.shape
    s12 = zeros_f64((2, c), y.device)
    s1, s2 = s12[0], s12[1]
    check(_lib.load().slu_bn_bwd_reduce(dz.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), n, c, h * w, s1.data_ptr(),
                                        s2.data_ptr(), _stream()), "slu_bn_bwd_reduce")
    return s1, s2


def bn_coeffs_fwd(sums, count: float, gamma, beta, eps: float, momentum: float, running_mean, running_var, train: bool):
    """(mean, invstd, a, b) [C] fp32 in one launch; in train mode the running statistics are updated in place."""
    c = gamma.numel()
    dev = gamma.device
    out = torch.empty((4, c), dtype=torch.float32, device=dev)
    s, q = (sums if sums is not None else (None, None))
    for t, nme in ((gamma, "gamma"), (beta, "beta")):
        _req(t, nme)
    if running_mean is not None:
        _req(running_mean, "running_mean")
        _req(running_var, "running_var")
    check(_lib.load().slu_bn_coeffs_fwd(_ptr(s), _ptr(q), float(count), gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum),
                                        1 if train else 0, _ptr(running_mean), _ptr(running_var), c, out[0].data_ptr(), out[1].data_ptr(),
                                        out[2].data_ptr(), out[3].data_ptr(), _stream()), "slu_bn_coeffs_fwd")
    if train and running_mean is not None:
        # the kernel wrote the running statistics through raw pointers: tell torch, so that caches keyed on tensor versions
        # (salsanext._tkey: folded eval-BatchNorm coefficients) see the change
        torch.autograd.graph.increment_version(running_mean)
        torch.autograd.graph.increment_version(running_var)
    return out[0], out[1], out[2], out[3]


def bn_coeffs_bwd(s1, s2, count: float, gamma, mean, invstd, train: bool):
    """(k1, k2, k3, dgamma, dbeta) [C] fp32 in one launch."""
    c = gamma.numel()
    out = torch.empty((5, c), dtype=torch.float32, device=gamma.device)
    check(_lib.load().slu_bn_coeffs_bwd(s1.data_ptr(), s2.data_ptr(), float(count), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                        1 if train else 0, c, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                        out[4].data_ptr(), _stream()), "slu_bn_coeffs_bwd")
    return out[0], out[1], out[2], out[3], out[4]


def affine(y, a=None, b=None, resid=None):
    """z = a[c]*y + b[c] + resid."""
    _req(y, "y")
    n, c, h, w = y.shape
    for t, nme in ((a, "a"), (b, "b")):
        if t is not None:
            _req(t, nme)
            if t.numel() != c:
                raise RuntimeError(f"{nme}: expected {c} elements")
    if resid is not None:
        _req(resid, "resid")
        if resid.shape != y.shape:
            raise RuntimeError("resid shape mismatch")
    z = torch.empty_like(y)
    check(_lib.load().slu_affine_fwd(y.data_ptr(), _ptr(a), _ptr(b), _ptr(resid), z.data_ptr(), n, c, h * w, _stream()), "slu_affine_fwd")
    return z


def bn_apply_fwd(y, sums, count: float, gamma, beta, eps: float, momentum: float, running_mean, running_var, train: bool, resid=None):
    """(z, mean, invstd): BatchNorm2d on y [+ resid] in ONE launch (coefficients derived inside; running statistics updated in train mode)."""
    _req(y, "y")
    n, c, h, w = y.shape
    for t, nme in ((gamma, "gamma"), (beta, "beta")):
        _req(t, nme)
    if resid is not None:
        _req(resid, "resid")
        if resid.shape != y.shape:
            raise RuntimeError("resid shape mismatch")
    if running_mean is not None:
        _req(running_mean, "running_mean")
        _req(running_var, "running_var")
    s, q = (sums if sums is not None else (None, None))
    z = torch.empty_like(y)
    mi = torch.empty((2, c), dtype=torch.float32, device=y.device)
    check(_lib.load().slu_bn_apply_fwd(y.data_ptr(), _ptr(s), _ptr(q), float(count), gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum),
                                       1 if train else 0, _ptr(running_mean), _ptr(running_var), _ptr(resid), z.data_ptr(), n, c, h * w,
                                       mi[0].data_ptr(), mi[1].data_ptr(), _stream()), "slu_bn_apply_fwd")
    if train and running_mean is not None:
        torch.autograd.graph.increment_version(running_mean)      # written through raw pointers (see bn_coeffs_fwd)
        torch.autograd.graph.increment_version(running_var)
    return z, mi[0], mi[1]


def bn_act_bwd(dz, y, s1=None, s2=None, count: float = 1.0, gamma=None, mean=None, invstd=None, train: bool = False, slope=None,
               want_dbias: bool = True):
    """(da, dbias f32 or None, dgamma, dbeta): the BatchNorm + LeakyReLU backward of a conv layer in ONE launch from the reduction sums s1 / s2
    (None: no BatchNorm), the bias gradient rounded to fp32 inside."""
    _req(dz, "dz")
    n, c, h, w = dz.shape
    has_bn = s1 is not None
    if y is not None:
        _req(y, "y")
        if y.shape != dz.shape:
            raise RuntimeError("y shape mismatch")
    da = torch.empty_like(dz)
    out = torch.empty((3, c), dtype=torch.float32, device=dz.device)            # dbias | dgamma | dbeta
    scratch = zeros_f64((2, c), dz.device) if want_dbias else None          # row 0: fp64 sums, row 1: the channels' ticket counters (uint32)
    check(_lib.load().slu_bn_act_bwd(dz.data_ptr(), _ptr(y), _ptr(s1), _ptr(s2), float(count), _ptr(gamma), _ptr(mean), _ptr(invstd),
                                     1 if has_bn else 0, 1 if train else 0, 0.0 if slope is None else float(slope), 0 if slope is None else 1,
                                     n, c, h * w, da.data_ptr(), None if scratch is None else scratch[0].data_ptr(),
                                     None if scratch is None else scratch[1].data_ptr(), out[0].data_ptr() if want_dbias else None,
                                     out[1].data_ptr(), out[2].data_ptr(), _stream()), "slu_bn_act_bwd")
    return da, (out[0] if want_dbias else None), (out[1] if has_bn else None), (out[2] if has_bn else None)


def act_affine_bwd(dz, y=None, k1=None, k2=None, k3=None, slope=None, want_dbias=True):
    """da = (k1*dz + k2 + k3*y) * leaky'(y);  dbias f64[C] = sum da."""
    _req(dz, "dz")
    n, c, h, w = dz.shape
    if y is not None:
        _req(y, "y")
        if y.shape != dz.shape:
            raise RuntimeError("y shape mismatch")
    for t, nme in ((k1, "k1"), (k2, "k2"), (k3, "k3")):
        if t is not None:
            _req(t, nme)
            if t.numel() != c:
                raise RuntimeError(f"{nme}: expected {c} elements")
    da = torch.empty_like(dz)
    db = zeros_f64((c,), dz.device) if want_dbias else None
    check(_lib.load().slu_act_affine_bwd(dz.data_ptr(), _ptr(y), _ptr(k1), _ptr(k2), _ptr(k3), 0.0 if slope is None else float(slope),
                                         0 if slope is None else 1, n, c, h * w, da.data_ptr(), _ptr(db), _stream()), "slu_act_affine_bwd")
    return da, db


def nchw_to_nhwc(x):
    _req(x, "x")
    n, c, h, w = x.shape
    cp = (c + 31) // 32 * 32
    out = torch.empty((n, h * w, cp), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_nchw_to_nhwc(x.data_ptr(), n, c, h * w, out.data_ptr(), _stream()), "slu_nchw_to_nhwc")
    return out


def gather_nhwc(srcs: Sequence[ConvSource]):
    arr = (_lib.ConvSrc * _lib.MAX_SRC)()
    n, h, w = _fill_srcs(arr, srcs)
    cin = sum(s.tensor.shape[1] // 4 if s.pixel_shuffle else s.tensor.shape[1] for s in srcs)
    cp = (cin + 31) // 32 * 32
    out = torch.empty((n, h * w, cp), dtype=torch.float32, device=srcs[0].tensor.device)
    check(_lib.load().slu_gather_nhwc(arr, len(srcs), n, h, w, out.data_ptr(), _stream()), "slu_gather_nhwc")
    return out


def split_grad(dcat, cbeg: int, src_shape, pixel_shuffle: bool, scale=None):
    _req(dcat, "dcat")
    n, ccat, h, w = dcat.shape
    sn, sc, sh, sw = src_shape
    if sn != n or (sh, sw) != ((h // 2, w // 2) if pixel_shuffle else (h, w)):
        raise RuntimeError("split_grad: source shape does not match the concatenated gradient")
    if scale is not None:
        _req(scale, "scale")
    out = torch.empty(tuple(src_shape), dtype=torch.float32, device=dcat.device)
    check(_lib.load().slu_split_grad(dcat.data_ptr(), n, ccat, int(cbeg), h, w, sc, 1 if pixel_shuffle else 0, _ptr(scale), out.data_ptr(),
                                     _stream()), "slu_split_grad")
    return out


def avgpool3s2_bwd(dy, scale, x_shape):
    _req(dy, "dy")
    n, c, h, w = x_shape
    if tuple(dy.shape) != (n, c, (h + 1) // 2, (w + 1) // 2):
        raise RuntimeError("avgpool3s2_bwd: dy shape mismatch")
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    check(_lib.load().slu_avgpool3s2_bwd(dy.data_ptr(), _ptr(scale), dx.data_ptr(), n, c, h, w, _stream()), "slu_avgpool3s2_bwd")
    return dx


def dgrad_weight(weight):
    """[Cout,Cin,k,k] -> [Cin,Cout,k,k] with taps mirrored: the weights of the data-gradient conv."""
    _req(weight, "weight")
    cout, cin, k, _ = weight.shape
    wd = torch.empty((cin, cout, k, k), dtype=torch.float32, device=weight.device)
    check(_lib.load().slu_dgrad_weight(weight.data_ptr(), cout, cin, k, wd.data_ptr(), _stream()), "slu_dgrad_weight")
    return wd


def conv2d_wgrad(da_t, in_t, n, h, w, cout, cin, ksize, dil, pad):
    """dW [cout,cin,k,k] from channel-last da_t [N,HW,Cop] and in_t [N,HW,Cip]."""
    _req(da_t, "da_t")
    _req(in_t, "in_t")
    cop, cip = (cout + 31) // 32 * 32, (cin + 31) // 32 * 32
    if tuple(da_t.shape) != (n, h * w, cop) or tuple(in_t.shape) != (n, h * w, cip):
        raise RuntimeError("conv2d_wgrad: operand shapes do not match")
    if w % 2:
        raise RuntimeError("conv2d_wgrad: W must be even")
    lib = _lib.load()
    scratch = torch.empty(lib.slu_wgrad_packed_floats(cout, cin, ksize), dtype=torch.float32, device=da_t.device)
    dw = torch.empty((cout, cin, ksize, ksize), dtype=torch.float32, device=da_t.device)
    check(lib.slu_conv2d_wgrad(da_t.data_ptr(), in_t.data_ptr(), n, h, w, cout, cin, ksize, dil, pad, scratch.data_ptr(), dw.data_ptr(),
                               _stream()), "slu_conv2d_wgrad")
    return dw


def conv2d_wgrad_nchw(da: torch.Tensor, srcs: Sequence[ConvSource], ksize: int, dil: int, pad: int):
    """dW [Cout, Cin, k, k] of a 3x3 (dil 1 / 2) or 2x2-dilated conv straight from NCHW da and its sources (PixelShuffle and multipliers
    included), or None when the shape is not covered (then: conv2d_wgrad on channel-last copies)."""
    _req(da, "da")
    n, cout, h, w = da.shape
    if (ksize, dil, pad) not in ((3, 1, 1), (3, 2, 2), (2, 2, 1)) or w % 16:
        return None
    if any(s.nbatch or s.cuse for s in srcs) or (h % 2 and any(s.pixel_shuffle for s in srcs)):
        return None
    arr = (_lib.ConvSrc * _lib.MAX_SRC)()
    if _fill_srcs(arr, srcs) != (n, h, w):
        raise RuntimeError("conv2d_wgrad_nchw: da and the sources disagree on (N, H, W)")
    cin = sum(s.tensor.shape[1] // 4 if s.pixel_shuffle else s.tensor.shape[1] for s in srcs)
    lib = _lib.load()
    scratch = _ZEROS32.take(lib.slu_wgrad_packed_floats(cout, cin, ksize), da.device)      # zero already: the launcher skips its fill
    dw = torch.empty((cout, cin, ksize, ksize), dtype=torch.float32, device=da.device)
    check(lib.slu_conv2d_wgrad_nchw(da.data_ptr(), arr, len(srcs), n, h, w, cout, ksize, dil, pad, scratch.data_ptr(), dw.data_ptr(), 1, _stream()),
          "slu_conv2d_wgrad_nchw")
    return dw


def conv1x1_wgrad_nchw(da: torch.Tensor, srcs: Sequence[ConvSource]):
    """dW [Cout, Cin, 1, 1] of a 1x1 conv straight from NCHW da and its plain sources, or None when the shape is not covered
    (then: conv2d_wgrad on channel-last copies)."""
    _req(da, "da")
    n, cout, h, w = da.shape
    if (h * w) % 32 or any(s.pixel_shuffle or s.scale is not None or s.nbatch or s.cuse for s in srcs):
        return None
    if any(s.tensor.shape[1] % 32 for s in srcs[:-1]) or any(tuple(s.tensor.shape[2:]) != (h, w) or s.tensor.shape[0] != n for s in srcs):
        return None
    arr = (_lib.ConvSrc * _lib.MAX_SRC)()
    _fill_srcs(arr, srcs)
    cin = sum(s.tensor.shape[1] for s in srcs)
    dw = _ZEROS32.take(cout * cin, da.device).view(cout, cin, 1, 1)                         # the kernel adds its partial sums into it
    check(_lib.load().slu_conv1x1_wgrad_nchw(da.data_ptr(), arr, len(srcs), n, h * w, cout, dw.data_ptr(), 1, _stream()), "slu_conv1x1_wgrad_nchw")
    return dw


# ------------------------------------------------------------------------------------------------
# ResNet-FPN data movement
# ------------------------------------------------------------------------------------------------
def maxpool3s2(x):
    _req(x, "x")
    n, c, h, w = x.shape
    y = torch.empty((n, c, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_maxpool3s2_fwd(x.data_ptr(), y.data_ptr(), n, c, h, w, _stream()), "slu_maxpool3s2_fwd")
    return y


def nearest_down(x, factor: int):
    _req(x, "x")
    n, c, h, w = x.shape
    if h % factor or w % factor:
        raise RuntimeError("nearest_down: H and W must be multiples of the factor")
    y = torch.empty((n, c, h // factor, w // factor), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_nearest_down(x.data_ptr(), y.data_ptr(), n, c, h, w, int(factor), _stream()), "slu_nearest_down")
    return y


def space_to_depth2(x):
    """[N,C,H,W] -> [N,4C,H/2,W/2], channel (2p+q)*C + c = x[c, 2y+p, 2x+q]."""
    _req(x, "x")
    n, c, h, w = x.shape
    if h % 2 or w % 2:
        raise RuntimeError("space_to_depth2: H and W must be even")
    y = torch.empty((n, 4 * c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_space_to_depth2(x.data_ptr(), y.data_ptr(), n, c, h, w, _stream()), "slu_space_to_depth2")
    return y


def space_to_depth2_cat(a, ca: int, b):
    """space_to_depth2 of cat(a[:, :ca], b) without materialising the concatenation."""
    _req(a, "a")
    _req(b, "b")
    n, ca_full, h, w = a.shape
    if b.shape[0] != n or tuple(b.shape[2:]) != (h, w) or not 0 < ca <= ca_full or h % 2 or w % 2:
        raise RuntimeError("space_to_depth2_cat: shape mismatch")
    cb = b.shape[1]
    y = torch.empty((n, 4 * (ca + cb), h // 2, w // 2), dtype=torch.float32, device=a.device)
    check(_lib.load().slu_space_to_depth2_cat(a.data_ptr(), ca_full, int(ca), b.data_ptr(), cb, y.data_ptr(), n, h, w, _stream()),
          "slu_space_to_depth2_cat")
    return y


def depth_to_space(x, r: int, elu_plus_one: bool = False, out: Optional[torch.Tensor] = None, c_off: int = 0):
    """nn.PixelShuffle(r), optionally followed by ELU(v) + 1.  With `out` [N,Ctot,H*r,W*r] the result is written into its
    channels [c_off, c_off + C/r^2) (several up-sampled maps sharing one concatenated buffer)."""
    _req(x, "x")
    n, c, h, w = x.shape
    if c % (r * r):
        raise RuntimeError("depth_to_space: C must be a multiple of r*r")
    cout = c // (r * r)
    if out is None:
        out = torch.empty((n, cout, h * r, w * r), dtype=torch.float32, device=x.device)
    else:
        _req(out, "out")
        if out.shape[0] != n or tuple(out.shape[2:]) != (h * r, w * r) or c_off < 0 or c_off + cout > out.shape[1]:
            raise RuntimeError("depth_to_space: `out` does not fit")
    check(_lib.load().slu_depth_to_space(x.data_ptr(), out.data_ptr(), n, cout, h, w, int(r), 1 if elu_plus_one else 0, int(c_off),
                                         out.shape[1], _stream()), "slu_depth_to_space")
    return out


def row_softmax_mul(score, value):
    """value * softmax(score, dim=-1); score [N,1,H,W], value [N,C,H,W]."""
    _req(score, "score")
    _req(value, "value")
    n, c, h, w = value.shape
    if tuple(score.shape) != (n, 1, h, w):
        raise RuntimeError("row_softmax_mul: score must be [N,1,H,W]")
    out = torch.empty_like(value)
    check(_lib.load().slu_row_softmax_mul(score.data_ptr(), value.data_ptr(), out.data_ptr(), n, c, h, w, _stream()), "slu_row_softmax_mul")
    return out


# ------------------------------------------------------------------------------------------------
# AUROC of error detection (metrics/auroc.py of the reference)
def bilinear_upsample(x: torch.Tensor, scale: int) -> torch.Tensor:
    """F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=False) on the device."""
    _req(x, "x")
    if x.dim() != 4 or int(scale) < 1:
        raise RuntimeError("bilinear_upsample: NCHW tensor and an integer scale >= 1 expected")
    n, c, h, w = x.shape
    y = torch.empty((n, c, h * int(scale), w * int(scale)), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_bilinear_upsample(x.data_ptr(), y.data_ptr(), n, c, h, w, int(scale), _stream()), "slu_bilinear_upsample")
    return y


def groupnorm(x: torch.Tensor, groups: int, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float = 1e-5, relu: bool = False,
              inplace: bool = False, return_stats: bool = False):
    """nn.GroupNorm(groups, C, eps)(x) [+ ReLU] for NCHW x."""
    _req(x, "x")
    if x.dim() != 4 or x.shape[1] % int(groups):
        raise RuntimeError(f"groupnorm: NCHW input with C divisible by groups={groups} expected, got {tuple(x.shape)}")
    n, c, h, w = x.shape
    for t, nme in ((gamma, "gamma"), (beta, "beta")):
        if t is not None:
            _req(t, nme)
            if t.numel() != c:
                raise RuntimeError(f"{nme}: expected {c} elements")
    stats = torch.empty((2, n * int(groups)), dtype=torch.float32, device=x.device)
    y = x if inplace else torch.empty_like(x)
    check(_lib.load().slu_groupnorm_fwd(x.data_ptr(), _ptr(gamma), _ptr(beta), n, c, h * w, int(groups), float(eps), 1 if relu else 0, stats[0].data_ptr(),
                                        stats[1].data_ptr(), y.data_ptr(), _stream()), "slu_groupnorm_fwd")
    return (y, stats) if return_stats else y


def spatial_softmax_gate(x: torch.Tensor, score: torch.Tensor, return_stats: bool = False):
    """x * softmax(score over H*W) + x  (SpatialAttention of semanticFCN_opt): x [N,C,H,W], score [N,1,H,W]."""
    _req(x, "x")
    _req(score, "score")
    if x.dim() != 4 or tuple(score.shape) != (x.shape[0], 1, x.shape[2], x.shape[3]):
        raise RuntimeError(f"spatial_softmax_gate: x [N,C,H,W] / score [N,1,H,W] expected, got {tuple(x.shape)} / {tuple(score.shape)}")
    n, c, h, w = x.shape
    stats = torch.empty(2 * n, dtype=torch.float32, device=x.device)
    out = torch.empty_like(x)
    check(_lib.load().slu_spatial_softmax_gate(x.data_ptr(), score.data_ptr(), stats.data_ptr(), out.data_ptr(), n, c, h * w, _stream()),
          "slu_spatial_softmax_gate")
    return (out, stats) if return_stats else out


# ------------------------------------------------------------------------------------------------
# training path of the FPN models (csrc/fpn_train.hip): forward / backward pieces of the autograd nodes in fpn_autograd.py
# ------------------------------------------------------------------------------------------------
POINTWISE_OPS = {"leaky": 0, "tanh": 1, "elu+1": 2, "silu": 3}


def pointwise_fwd(x: torch.Tensor, op: str, slope: float = 0.0) -> torch.Tensor:
    """LeakyReLU(slope) (0 = ReLU) / tanh / ELU + 1 / SiLU, elementwise."""
    _req(x, "x")
    y = torch.empty_like(x)
    if x.numel():
        check(_lib.load().slu_pointwise_fwd(x.data_ptr(), y.data_ptr(), x.numel(), POINTWISE_OPS[op], float(slope), _stream()), "slu_pointwise_fwd")
    return y


def pointwise_bwd(dy: torch.Tensor, y: torch.Tensor, op: str, slope: float = 0.0) -> torch.Tensor:
    """dx = dy * f'(x), from the forward's output y (op 'silu': from the forward's INPUT, passed as `y`)."""
    _req(dy, "dy")
    _req(y, "y")
    if dy.shape != y.shape:
        raise RuntimeError("pointwise_bwd: shape mismatch")
    dx = torch.empty_like(dy)
    if dy.numel():
        check(_lib.load().slu_pointwise_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), POINTWISE_OPS[op], float(slope), _stream()),
              "slu_pointwise_bwd")
    return dx


def maxpool3s2_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _req(x, "x")
    _req(dy, "dy")
    n, c, h, w = x.shape
    if tuple(dy.shape) != (n, c, (h + 1) // 2, (w + 1) // 2):
        raise RuntimeError("maxpool3s2_bwd: dy does not match the pooled shape")
    dx = torch.empty_like(x)
    check(_lib.load().slu_maxpool3s2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, c, h, w, _stream()), "slu_maxpool3s2_bwd")
    return dx


def nearest_down_bwd(dy: torch.Tensor, factor: int) -> torch.Tensor:
    _req(dy, "dy")
    n, c, oh, ow = dy.shape
    dx = torch.empty((n, c, oh * int(factor), ow * int(factor)), dtype=torch.float32, device=dy.device)
    check(_lib.load().slu_nearest_down_bwd(dy.data_ptr(), dx.data_ptr(), n, c, oh * int(factor), ow * int(factor), int(factor), _stream()),
          "slu_nearest_down_bwd")
    return dx


def replace_tail(x: torch.Tensor, meta: torch.Tensor) -> torch.Tensor:
    """cat(x[:, :C-m], meta) with m = meta.shape[1]."""
    _req(x, "x")
    _req(meta, "meta")
    n, c, h, w = x.shape
    m = meta.shape[1]
    if tuple(meta.shape) != (n, m, h, w) or not 0 < m <= c:
        raise RuntimeError(f"replace_tail: x {tuple(x.shape)} / meta {tuple(meta.shape)}")
    out = torch.empty_like(x)
    check(_lib.load().slu_replace_tail_fwd(x.data_ptr(), meta.data_ptr(), out.data_ptr(), n, c, m, h, w, _stream()), "slu_replace_tail_fwd")
    return out


def replace_tail_bwd(dout: torch.Tensor, m: int, want_dx: bool = True, want_dmeta: bool = True):
    _req(dout, "dout")
    n, c, h, w = dout.shape
    dx = torch.empty_like(dout) if want_dx else None
    dmeta = torch.empty((n, m, h, w), dtype=torch.float32, device=dout.device) if want_dmeta else None
    if want_dx or want_dmeta:
        check(_lib.load().slu_replace_tail_bwd(dout.data_ptr(), _ptr(dx), _ptr(dmeta), n, c, int(m), h, w, _stream()), "slu_replace_tail_bwd")
    return dx, dmeta


def row_softmax_mul_bwd(score, value, dout, want_dscore: bool = True, want_dvalue: bool = True):
    _req(score, "score")
    _req(value, "value")
    _req(dout, "dout")
    n, c, h, w = value.shape
    if tuple(score.shape) != (n, 1, h, w) or dout.shape != value.shape:
        raise RuntimeError("row_softmax_mul_bwd: shape mismatch")
    dscore = torch.empty_like(score) if want_dscore else None
    dvalue = torch.empty_like(value) if want_dvalue else None
    if want_dscore or want_dvalue:
        check(_lib.load().slu_row_softmax_mul_bwd(score.data_ptr(), value.data_ptr(), dout.data_ptr(), _ptr(dscore), _ptr(dvalue), n, c, h, w, _stream()),
              "slu_row_softmax_mul_bwd")
    return dscore, dvalue


def depth_to_space_bwd(dy: torch.Tensor, cout: int, r: int, c_off: int = 0) -> torch.Tensor:
    """Gradient of depth_to_space w.r.t. its input from the channel slice [c_off, c_off + cout) of dy [N,Ctot,H*r,W*r]."""
    _req(dy, "dy")
    n, ctot, oh, ow = dy.shape
    if oh % r or ow % r or c_off < 0 or c_off + cout > ctot:
        raise RuntimeError("depth_to_space_bwd: shape mismatch")
    dx = torch.empty((n, cout * r * r, oh // r, ow // r), dtype=torch.float32, device=dy.device)
    check(_lib.load().slu_depth_to_space_bwd(dy.data_ptr(), dx.data_ptr(), n, int(cout), oh // r, ow // r, int(r), int(c_off), ctot, _stream()),
          "slu_depth_to_space_bwd")
    return dx


def bilinear_upsample_bwd(dy: torch.Tensor, scale: int) -> torch.Tensor:
    _req(dy, "dy")
    n, c, oh, ow = dy.shape
    if oh % scale or ow % scale:
        raise RuntimeError("bilinear_upsample_bwd: shape mismatch")
    dx = torch.empty((n, c, oh // scale, ow // scale), dtype=torch.float32, device=dy.device)
    check(_lib.load().slu_bilinear_upsample_bwd(dy.data_ptr(), dx.data_ptr(), n, c, oh // scale, ow // scale, int(scale), _stream()),
          "slu_bilinear_upsample_bwd")
    return dx


def groupnorm_bwd(x, y, dy, gamma, stats, groups: int, relu: bool, want_affine: bool = True):
    """(dx, dgamma f64[C] | None, dbeta f64[C] | None); stats = the [2, N*groups] (mean, rstd) tensor of groupnorm(return_stats=True)."""
    _req(x, "x")
    _req(dy, "dy")
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    dg = zeros_f64((c,), x.device) if want_affine else None
    db = zeros_f64((c,), x.device) if want_affine else None
    check(_lib.load().slu_groupnorm_bwd(x.data_ptr(), _ptr(y), dy.data_ptr(), _ptr(gamma), stats[0].data_ptr(), stats[1].data_ptr(), dx.data_ptr(),
                                        _ptr(dg), _ptr(db), n, c, h * w, int(groups), 1 if relu else 0, _stream()), "slu_groupnorm_bwd")
    return dx, dg, db


def spatial_softmax_gate_bwd(x, score, stats, dout):
    _req(x, "x")
    _req(score, "score")
    _req(dout, "dout")
    n, c, h, w = x.shape
    dx, dscore = torch.empty_like(x), torch.empty_like(score)
    ws = torch.empty((n, h * w), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_spatial_softmax_gate_bwd(x.data_ptr(), score.data_ptr(), stats.data_ptr(), dout.data_ptr(), dx.data_ptr(), dscore.data_ptr(),
                                                   ws.data_ptr(), n, c, h * w, _stream()), "slu_spatial_softmax_gate_bwd")
    return dx, dscore


# ------------------------------------------------------------------------------------------------
AUROC_MODES = {"alpha": 0, "logits": 1, "probs": 2}
AUROC_SCORES = {"entropy": 0, "entropy_norm": 1, "mi": 2, "mi_norm": 3, "1-maxprob": 4}


def auroc_scores(preds: torch.Tensor, labels: torch.Tensor, mode: str, score: str, ignore_index=None, eps: float = 1e-12,
                 score_override: Optional[torch.Tensor] = None):
    """(score[B,H,W] fp32, flag[B,H,W] uint8: 0 correct / 1 error / 2 ignored) -- slu_auroc_scores."""
    _req(preds, "preds")
    _req(labels, "labels", torch.int64)
    if preds.dim() != 4 or tuple(labels.shape) != (preds.shape[0], preds.shape[2], preds.shape[3]):
        raise RuntimeError(f"auroc_scores: preds [B,C,H,W] / labels [B,H,W] expected, got {tuple(preds.shape)} / {tuple(labels.shape)}")
    if mode not in AUROC_MODES or score not in AUROC_SCORES:
        raise ValueError(f"auroc_scores: unknown mode {mode!r} or score {score!r}")
    b, c, h, w = preds.shape
    if score_override is not None:
        _req(score_override, "score_override")
        if tuple(score_override.shape) != (b, h, w):
            raise RuntimeError(f"score_override: expected {(b, h, w)}, got {tuple(score_override.shape)}")
    s = torch.empty((b, h, w), dtype=torch.float32, device=preds.device)
    f = torch.empty((b, h, w), dtype=torch.uint8, device=preds.device)
    check(_lib.load().slu_auroc_scores(preds.data_ptr(), labels.data_ptr(), _ptr(score_override), b, c, h * w, AUROC_MODES[mode], AUROC_SCORES[score],
                                       0 if ignore_index is None else 1, 0 if ignore_index is None else int(ignore_index), float(eps),
                                       s.data_ptr(), f.data_ptr(), _stream()), "slu_auroc_scores")
    return s, f


def auroc_from_samples(scores: torch.Tensor, is_error: torch.Tensor, want_sorted: bool = False):
    """AUROC of 1-D device samples (score fp32, is_error uint8 in {0,1}) -> (auroc, P, N[, sorted_scores, sorted_is_error])."""
    _req(scores, "scores")
    _req(is_error, "is_error", torch.uint8)
    if scores.dim() != 1 or scores.shape != is_error.shape or scores.numel() == 0:
        raise RuntimeError("auroc_from_samples: two equally long, non-empty 1-D tensors expected")
    lib = _lib.load()
    n = scores.numel()
    ws = torch.empty(lib.slu_auroc_workspace_bytes(n), dtype=torch.uint8, device=scores.device)
    out = torch.empty(3, dtype=torch.float64, device=scores.device)
    ss = torch.empty_like(scores) if want_sorted else None
    se = torch.empty_like(is_error) if want_sorted else None
    check(lib.slu_auroc_compute(scores.data_ptr(), is_error.data_ptr(), n, ws.data_ptr(), ws.numel(), out.data_ptr(), _ptr(ss), _ptr(se), _stream()),
          "slu_auroc_compute")
    a, p, nn_ = (float(v) for v in out.cpu())
    return (a, int(p), int(nn_), ss, se) if want_sorted else (a, int(p), int(nn_))


# ------------------------------------------------------------------------------------------------
# accuracy vs uncertainty bins (models/evaluator.py UncertaintyAccuracyAggregator)
# ------------------------------------------------------------------------------------------------
def ua_samples(labels: torch.Tensor, preds: torch.Tensor, uncertainty: torch.Tensor, ignore_ids=()):
    """(u fp32 [n] clamped to [0,1], flag uint8 [n]: 1 correct / 0 wrong / 2 label in ignore_ids), flattened scan order."""
    _req(labels, "labels", torch.int64)
    _req(preds, "preds", torch.int64)
    _req(uncertainty, "uncertainty")
    if not (labels.shape == preds.shape == uncertainty.shape) or labels.numel() == 0:
        raise RuntimeError("ua_samples: labels, preds and uncertainty must have the same non-empty shape")
    n = labels.numel()
    ids = torch.tensor(list(ignore_ids), dtype=torch.int64, device=labels.device) if len(ignore_ids) else None
    u = torch.empty(n, dtype=torch.float32, device=labels.device)
    f = torch.empty(n, dtype=torch.uint8, device=labels.device)
    check(_lib.load().slu_ua_samples(labels.data_ptr(), preds.data_ptr(), uncertainty.data_ptr(), n, _ptr(ids), 0 if ids is None else ids.numel(),
                                     u.data_ptr(), f.data_ptr(), _stream()), "slu_ua_samples")
    return u, f


def binned_counts(u: torch.Tensor, correct: torch.Tensor, edges: torch.Tensor):
    """np.histogram(u, bins=edges) and the histogram of the correct ones, on the device: (count int64 [K], n_correct int64 [K])."""
    _req(u, "u")
    _req(correct, "correct", torch.uint8)
    _req(edges, "edges")
    if u.dim() != 1 or u.shape != correct.shape or u.numel() == 0 or edges.dim() != 1 or not 2 <= edges.numel() <= 257:
        raise RuntimeError("binned_counts: 1-D samples and 2..257 edges expected")
    k = edges.numel() - 1
    cnt = torch.zeros(k, dtype=torch.int64, device=u.device)
    ok = torch.zeros(k, dtype=torch.int64, device=u.device)
    check(_lib.load().slu_binned_counts(u.data_ptr(), correct.data_ptr(), u.numel(), edges.data_ptr(), k, cnt.data_ptr(), ok.data_ptr(), _stream()),
          "slu_binned_counts")
    return cnt, ok


def ece_samples(preds: torch.Tensor, labels: torch.Tensor, mode: str, ignore_index=None, eps: float = 1e-12):
    """(conf fp32 [B,H,W] in [0,1], flag uint8 [B,H,W]: 1 correct / 0 wrong / 2 ignored) -- slu_ece_samples (metrics/ece.py:55-84)."""
    _req(preds, "preds")
    _req(labels, "labels", torch.int64)
    if preds.dim() != 4 or tuple(labels.shape) != (preds.shape[0], preds.shape[2], preds.shape[3]):
        raise RuntimeError(f"ece_samples: preds [B,C,H,W] / labels [B,H,W] expected, got {tuple(preds.shape)} / {tuple(labels.shape)}")
    if mode not in AUROC_MODES:
        raise ValueError(f"ece_samples: unknown mode {mode!r}")
    b, c, h, w = preds.shape
    conf = torch.empty((b, h, w), dtype=torch.float32, device=preds.device)
    flag = torch.empty((b, h, w), dtype=torch.uint8, device=preds.device)
    check(_lib.load().slu_ece_samples(preds.data_ptr(), labels.data_ptr(), b, c, h * w, AUROC_MODES[mode], 0 if ignore_index is None else 1,
                                      0 if ignore_index is None else int(ignore_index), float(eps), conf.data_ptr(), flag.data_ptr(), _stream()),
          "slu_ece_samples")
    return conf, flag


def binned_stats(u: torch.Tensor, correct: torch.Tensor, edges: torch.Tensor):
    """The three np.histogram calls of metrics/ece.py:136-140 on the device: (count int64 [K], n_correct int64 [K], sum_u float64 [K])."""
    _req(u, "u")
    _req(correct, "correct", torch.uint8)
    _req(edges, "edges")
    if u.dim() != 1 or u.shape != correct.shape or u.numel() == 0 or edges.dim() != 1 or not 2 <= edges.numel() <= 257:
        raise RuntimeError("binned_stats: 1-D samples and 2..257 edges expected")
    k = edges.numel() - 1
    cnt = torch.zeros(k, dtype=torch.int64, device=u.device)
    ok = torch.zeros(k, dtype=torch.int64, device=u.device)
    su = torch.zeros(k, dtype=torch.float64, device=u.device)
    check(_lib.load().slu_binned_stats(u.data_ptr(), correct.data_ptr(), u.numel(), edges.data_ptr(), k, cnt.data_ptr(), ok.data_ptr(), su.data_ptr(),
                                       _stream()), "slu_binned_stats")
    return cnt, ok, su


# ------------------------------------------------------------------------------------------------
# Tversky loss (models/losses.py TverskyLoss)
# ------------------------------------------------------------------------------------------------
TVERSKY_ACTS = {"logits": 0, "probs": 1, "log_probs": 2}
TVERSKY_REDUCTIONS = {"mean": 0, "sum": 1, "none": 2}


def tversky_fwd(x: torch.Tensor, labels: torch.Tensor, model_act: str, ignore_index, alpha: float, beta: float, smooth: float, reduction: str):
    """-> (loss [1] or [C], coef [2,C] for the backward, any_valid [1]) -- slu_tversky_fwd."""
    _req(x, "outputs")
    _req(labels, "labels", torch.int64)
    if x.dim() != 4 or tuple(labels.shape) != (x.shape[0], x.shape[2], x.shape[3]):
        raise RuntimeError(f"tversky: outputs [B,C,H,W] / labels [B,H,W] expected, got {tuple(x.shape)} / {tuple(labels.shape)}")
    b, c, h, w = x.shape
    red = TVERSKY_REDUCTIONS[reduction]
    sums = torch.empty(3 * c, dtype=torch.float64, device=x.device)
    coef = torch.empty((2, c), dtype=torch.float32, device=x.device)
    loss = torch.empty(c if red == 2 else 1, dtype=torch.float32, device=x.device)
    anyv = torch.empty(1, dtype=torch.float32, device=x.device)
    check(_lib.load().slu_tversky_fwd(x.data_ptr(), labels.data_ptr(), b, c, h * w, TVERSKY_ACTS[model_act], 0 if ignore_index is None else 1,
                                      0 if ignore_index is None else int(ignore_index), float(alpha), float(beta), float(smooth), red,
                                      sums.data_ptr(), coef.data_ptr(), loss.data_ptr(), anyv.data_ptr(), _stream()), "slu_tversky_fwd")
    return loss, coef, anyv


def tversky_bwd(x: torch.Tensor, labels: torch.Tensor, model_act: str, ignore_index, alpha: float, beta: float, coef: torch.Tensor,
                grad_out: torch.Tensor):
    _req(x, "outputs")
    _req(labels, "labels", torch.int64)
    _req(coef, "coef")
    _req(grad_out, "grad_out")
    b, c, h, w = x.shape
    if grad_out.numel() not in (1, c):
        raise RuntimeError("tversky_bwd: grad_out must have 1 or C elements")
    gx = torch.empty_like(x)
    check(_lib.load().slu_tversky_bwd(x.data_ptr(), labels.data_ptr(), b, c, h * w, TVERSKY_ACTS[model_act], 0 if ignore_index is None else 1,
                                      0 if ignore_index is None else int(ignore_index), float(alpha), float(beta), coef.data_ptr(),
                                      grad_out.data_ptr(), 1 if (grad_out.numel() == c and c > 1) else 0, gx.data_ptr(), _stream()), "slu_tversky_bwd")
    return gx


# ------------------------------------------------------------------------------------------------
# per-pixel Dirichlet losses (losses/dirichlet_losses.py, losses/regularizers.py)
# ------------------------------------------------------------------------------------------------
DIRICHLET_LOSS_KINDS = {"nll_dircat": 0, "digamma_ce": 1, "brier": 2, "mse": 3, "kl_off_uniform": 4, "complement_kl": 5, "wrong_low_evidence": 6,
                        "kl_off_uniform_weighted": 7}


def _dirichlet_loss_args(alpha, labels):
    _req(alpha, "alpha")
    _req(labels, "labels", torch.int64)
    if alpha.dim() != 4 or tuple(labels.shape) != (alpha.shape[0], alpha.shape[2], alpha.shape[3]):
        raise RuntimeError(f"dirichlet loss: alpha [B,C,H,W] / labels [B,H,W] expected, got {tuple(alpha.shape)} / {tuple(labels.shape)}")
    return alpha.shape


def dirichlet_loss_fwd(alpha: torch.Tensor, labels: torch.Tensor, kind: str, param: float, eps: float, ignore_index):
    """-> (sum float64 [1], count int64 [1]) over the valid pixels."""
    b, c, h, w = _dirichlet_loss_args(alpha, labels)
    s = torch.empty(1, dtype=torch.float64, device=alpha.device)
    n = torch.empty(1, dtype=torch.int64, device=alpha.device)
    check(_lib.load().slu_dirichlet_loss_fwd(alpha.data_ptr(), labels.data_ptr(), b, c, h * w, DIRICHLET_LOSS_KINDS[kind], float(param), float(eps),
                                             0 if ignore_index is None else 1, 0 if ignore_index is None else int(ignore_index), s.data_ptr(),
                                             n.data_ptr(), _stream()), "slu_dirichlet_loss_fwd")
    return s, n


def dirichlet_loss_bwd(alpha: torch.Tensor, labels: torch.Tensor, kind: str, param: float, eps: float, ignore_index, gscale: torch.Tensor):
    b, c, h, w = _dirichlet_loss_args(alpha, labels)
    _req(gscale, "gscale")
    g = torch.empty_like(alpha)
    check(_lib.load().slu_dirichlet_loss_bwd(alpha.data_ptr(), labels.data_ptr(), b, c, h * w, DIRICHLET_LOSS_KINDS[kind], float(param), float(eps),
                                             0 if ignore_index is None else 1, 0 if ignore_index is None else int(ignore_index),
                                             gscale.data_ptr(), g.data_ptr(), _stream()), "slu_dirichlet_loss_bwd")
    return g


def build_normals(xyz: torch.Tensor, norm_factor: float = 0.25):
    """xyz fp32 [H, W, C >= 3] on the GPU -> unit normals fp32 [H, W, 3] (Scharr derivatives, cross product, normalisation)."""
    _req(xyz, "xyz")
    if xyz.dim() != 3 or xyz.shape[2] < 3:
        raise RuntimeError(f"xyz: expected [H, W, C >= 3], got {tuple(xyz.shape)}")
    h, w, c = xyz.shape
    out = torch.empty((h, w, 3), dtype=torch.float32, device=xyz.device)
    check(_lib.load().slu_build_normals(xyz.data_ptr(), h, w, c, float(norm_factor), out.data_ptr(), _stream()), "slu_build_normals")
    return out


def group_by_class(labels: torch.Tensor, values: torch.Tensor, num_classes: int):
    """labels int64 [n], values fp32 [n] (device) -> (grouped fp32 [n]: class 0's samples, then class 1's, ... in scan order; the first
    counts.sum() entries are meaningful), counts int64 [C] (device)."""
    _req(labels, "labels", torch.int64)
    _req(values, "values")
    if labels.dim() != 1 or labels.shape != values.shape or labels.numel() == 0:
        raise RuntimeError(f"group_by_class: equally long non-empty 1-D labels / values expected, got {tuple(labels.shape)} / {tuple(values.shape)}")
    n = labels.numel()
    lib = _lib.load()
    ws = torch.empty(lib.slu_group_by_class_workspace_bytes(n), dtype=torch.uint8, device=labels.device)
    out = torch.empty(n, dtype=torch.float32, device=labels.device)
    counts = torch.empty(int(num_classes), dtype=torch.int64, device=labels.device)
    check(lib.slu_group_by_class(labels.data_ptr(), values.data_ptr(), n, int(num_classes), out.data_ptr(), counts.data_ptr(), ws.data_ptr(),
                                 ws.numel(), _stream()), "slu_group_by_class")
    return out, counts


def _c_params(params):
    import ctypes as C
    vals = [float(v) for v in params]
    return (C.c_float * max(1, len(vals)))(*vals), len(vals)


def dirichlet_loss_fwd_ex(alpha: torch.Tensor, labels: torch.Tensor, kind: str, params, eps: float, ignore_index):
    """-> (sums float64 [2] = {sum of values, sum of gates (kind "wrong_low_evidence")}, count int64 [1]); `params` per include/slu.h."""
    b, c, h, w = _dirichlet_loss_args(alpha, labels)
    s = torch.empty(2, dtype=torch.float64, device=alpha.device)
    n = torch.empty(1, dtype=torch.int64, device=alpha.device)
    arr, k = _c_params(params)
    check(_lib.load().slu_dirichlet_loss_fwd_ex(alpha.data_ptr(), labels.data_ptr(), b, c, h * w, DIRICHLET_LOSS_KINDS[kind], arr, k, float(eps),
                                                0 if ignore_index is None else 1, 0 if ignore_index is None else int(ignore_index), s.data_ptr(),
                                                n.data_ptr(), _stream()), "slu_dirichlet_loss_fwd_ex")
    return s, n


def dirichlet_loss_bwd_ex(alpha: torch.Tensor, labels: torch.Tensor, kind: str, params, eps: float, ignore_index, gscale: torch.Tensor):
    b, c, h, w = _dirichlet_loss_args(alpha, labels)
    _req(gscale, "gscale")
    g = torch.empty_like(alpha)
    arr, k = _c_params(params)
    check(_lib.load().slu_dirichlet_loss_bwd_ex(alpha.data_ptr(), labels.data_ptr(), b, c, h * w, DIRICHLET_LOSS_KINDS[kind], arr, k, float(eps),
                                                0 if ignore_index is None else 1, 0 if ignore_index is None else int(ignore_index),
                                                gscale.data_ptr(), g.data_ptr(), _stream()), "slu_dirichlet_loss_bwd_ex")
    return g


# ------------------------------------------------------------------------------------------------
# spherical projection (dataset/utils.py of the reference)
# ------------------------------------------------------------------------------------------------
def spherical_projection(pc: torch.Tensor, height: int, width: int, theta_range=None, bins_h: Optional[torch.Tensor] = None,
                         bins_increasing: bool = False, keep_farthest: bool = False, flip: bool = False, out: Optional[torch.Tensor] = None):
    """pc float64 [N, C >= 3] on the GPU -> (img fp32 [H, W, C], theta_range float64 [2] on the device).  Default: the nearest point of a
    pixel survives; keep_farthest = the reference's sort_largest_first=True; bins_h: float64 [H] explicit monotone row bins on the device;
    flip: columns reversed and y negated while writing (the dataloaders' flip augmentation)."""
    _req(pc, "pc", torch.float64)
    if pc.dim() != 2 or pc.shape[1] < 3 or pc.shape[0] == 0:
        raise RuntimeError(f"pc: expected [N, C >= 3], got {tuple(pc.shape)}")
    n, c = pc.shape
    lib = _lib.load()
    if bins_h is not None:
        _req(bins_h, "bins_h", torch.float64)
        if bins_h.dim() != 1 or bins_h.numel() != int(height):
            raise RuntimeError(f"bins_h: expected {int(height)} row bins, got {tuple(bins_h.shape)}")
    ws = torch.empty(lib.slu_spherical_projection_workspace_bytes(n, int(height), int(width)), dtype=torch.uint8, device=pc.device)
    if out is None:
        img = torch.empty((int(height), int(width), c), dtype=torch.float32, device=pc.device)
    else:
        img = _req(out, "out")
        if tuple(img.shape) != (int(height), int(width), c):
            raise RuntimeError(f"out: expected {(int(height), int(width), c)}, got {tuple(img.shape)}")
    tr = torch.empty(2, dtype=torch.float64, device=pc.device)
    use_data = theta_range is None
    tmin, tmax = (0.0, 0.0) if use_data else (float(theta_range[0]), float(theta_range[1]))
    check(lib.slu_spherical_projection_ex(pc.data_ptr(), n, c, int(height), int(width), 1 if use_data else 0, tmin, tmax, _ptr(bins_h),
                                          1 if bins_increasing else 0, 1 if keep_farthest else 0, 1 if flip else 0, ws.data_ptr(), ws.numel(),
                                          img.data_ptr(), tr.data_ptr(), _stream()), "slu_spherical_projection_ex")
    return img, tr


def resize_nearest_hwc(img: torch.Tensor, out_h: int, out_w: int, flip: bool = False) -> torch.Tensor:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_NEAREST) of an [H, W, C] fp32 image [+ the dataloader's flip afterwards]."""
    _req(img, "img")
    if img.dim() != 3:
        raise RuntimeError(f"resize_nearest_hwc: [H, W, C] expected, got {tuple(img.shape)}")
    h, w, c = img.shape
    out = torch.empty((int(out_h), int(out_w), c), dtype=torch.float32, device=img.device)
    check(_lib.load().slu_resize_nearest_hwc(img.data_ptr(), h, w, c, out.data_ptr(), int(out_h), int(out_w), 1 if flip else 0, _stream()),
          "slu_resize_nearest_hwc")
    return out


def kitti_decode(xyzi: torch.Tensor, label: torch.Tensor, lut: torch.Tensor, bad_count: torch.Tensor, rotate_deg: Optional[float] = None):
    """.bin contents fp32 [N, 4] + .label contents int32 [N] (the uint32 words) + id_map LUT int32 -> float64 [N, 5] (x, y, z, i, class);
    rotate_deg: yaw of dataset.utils.rotate_z.  bad_count int32 [1] accumulates labels without a LUT entry."""
    import math
    _req(xyzi, "xyzi")
    _req(label, "label", torch.int32)
    _req(lut, "lut", torch.int32)
    _req(bad_count, "bad_count", torch.int32)
    if xyzi.dim() != 2 or xyzi.shape[1] != 4 or label.dim() != 1 or label.numel() != xyzi.shape[0] or xyzi.shape[0] == 0:
        raise RuntimeError(f"kitti_decode: xyzi [N, 4] / label [N] expected, got {tuple(xyzi.shape)} / {tuple(label.shape)}")
    n = xyzi.shape[0]
    pc = torch.empty((n, 5), dtype=torch.float64, device=xyzi.device)
    ang = 0.0 if rotate_deg is None else math.radians(float(rotate_deg))
    if rotate_deg is not None:
        import numpy as np                      # the reference builds the matrix with numpy's cos / sin of np.radians(angle)
        ang = float(np.radians(float(rotate_deg)))
        ca, sa = float(np.cos(ang)), float(np.sin(ang))
    else:
        ca, sa = 1.0, 0.0
    check(_lib.load().slu_kitti_decode(xyzi.data_ptr(), label.data_ptr(), n, lut.data_ptr(), lut.numel(), 0 if rotate_deg is None else 1, ca, sa,
                                       pc.data_ptr(), bad_count.data_ptr(), _stream()), "slu_kitti_decode")
    return pc


def range_image_split(img: torch.Tensor, normals: Optional[torch.Tensor], range_out, refl_out, xyz_out, normals_out, labels_out) -> None:
    """Projected image fp32 [H, W, C >= 5] (+ normals [H, W, 3]) -> the dataloader's outputs written into the given (slices of batch)
    tensors: range [1,H,W], reflectivity [1,H,W], xyz [3,H,W], normals [3,H,W], labels int64 [1,H,W]."""
    _req(img, "img")
    h, w, c = img.shape
    for t, nme, shape, dt in ((range_out, "range_out", (1, h, w), torch.float32), (refl_out, "refl_out", (1, h, w), torch.float32),
                              (xyz_out, "xyz_out", (3, h, w), torch.float32), (labels_out, "labels_out", (1, h, w), torch.int64)):
        _req(t, nme, dt)
        if tuple(t.shape) != shape:
            raise RuntimeError(f"{nme}: expected {shape}, got {tuple(t.shape)}")
    if normals is not None:
        _req(normals, "normals")
        _req(normals_out, "normals_out")
        if tuple(normals.shape) != (h, w, 3) or tuple(normals_out.shape) != (3, h, w):
            raise RuntimeError("normals: expected [H, W, 3] in and [3, H, W] out")
    check(_lib.load().slu_range_image_split(img.data_ptr(), _ptr(normals), h, w, c, range_out.data_ptr(), refl_out.data_ptr(), xyz_out.data_ptr(),
                                            _ptr(normals_out) if normals is not None else None, labels_out.data_ptr(), _stream()), "slu_range_image_split")


# ------------------------------------------------------------------------------------------------
# Dropout2d multipliers of a whole MC evaluation in one launch (csrc/dropout_draw.hip)
# ------------------------------------------------------------------------------------------------
class DropoutPlan:
    """Device tables for slu_dropout_draw: `sites` = [(C, p, active)], `outs` = [(key, C, (site, offset) x <= 3, shuffled)] for batch size n."""

    def __init__(self, n: int, sites, outs, device):
        import numpy as np
        self.n, self.device = int(n), device
        st = np.zeros(len(sites), dtype=np.dtype([("begin", "<i8"), ("C", "<i4"), ("active", "<i4"), ("p", "<f4"), ("reserved", "<i4")]))
        g = 0
        for i, (c, p, active) in enumerate(sites):
            st[i] = (g, c, 1 if active else 0, p, 0)
            if active:
                g += self.n * c
        self.draws = g
        ot = np.zeros(len(outs), dtype=np.dtype([("begin", "<i8"), ("C", "<i4"), ("site_a", "<i4"), ("off_a", "<i4"), ("site_b", "<i4"), ("off_b", "<i4"),
                                                 ("site_c", "<i4"), ("off_c", "<i4"), ("shuffled", "<i4")]))
        e, self.slices = 0, []
        for i, (key, c, refs, shuffled) in enumerate(outs):
            refs = list(refs) + [(-1, 0)] * (3 - len(refs))
            ot[i] = (e, c, refs[0][0], refs[0][1], refs[1][0], refs[1][1], refs[2][0], refs[2][1], 1 if shuffled else 0)
            self.slices.append((key, e, c))
            e += self.n * c
        self.total = e
        assert st.dtype.itemsize == 24 and ot.dtype.itemsize == 40
        self.sites = torch.from_numpy(st.view(np.uint8).copy()).to(device)
        self.outs = torch.from_numpy(ot.view(np.uint8).copy()).to(device)
        self.nsites, self.nout = len(sites), len(outs)

    def run(self, out: Optional[torch.Tensor] = None):
        """{key: [n, C] multiplier}: one launch; consumes ceil(draws / 4) (rounded up to a multiple of 4) counters of torch's CUDA generator.
        out: a float32 buffer of `self.total` elements to draw into (a captured HIP graph reads the multipliers from fixed addresses)."""
        gen = torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()]
        seed, off = int(gen.initial_seed()), int(gen.get_offset())
        adv = ((self.draws + 3) // 4 + 3) // 4 * 4
        gen.set_offset(off + adv)
        if out is None:
            buf = torch.empty(self.total, dtype=torch.float32, device=self.device)
        else:
            buf = _req(out, "out")
            if buf.numel() != self.total:
                raise RuntimeError(f"DropoutPlan.run: out holds {buf.numel()} elements, the plan needs {self.total}")
        check(_lib.load().slu_dropout_draw(self.sites.data_ptr(), self.nsites, self.outs.data_ptr(), self.nout, self.n, seed & (2 ** 64 - 1), off, buf.data_ptr(),
                                           self.total, _stream()), "slu_dropout_draw")
        return {key: buf[b:b + self.n * c].view(self.n, c) for key, b, c in self.slices}


# ------------------------------------------------------------------------------------------------
# EfficientNetV2 pieces (csrc/effnet_ops.hip)
# ------------------------------------------------------------------------------------------------
def dwconv3x3(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, act: str = "none") -> torch.Tensor:
    """Depthwise 3x3 (padding 1, stride 1 / 2) + bias [+ SiLU]; w [C, 9] (eval BatchNorm already folded in)."""
    _req(x, "x")
    _req(w, "w")
    n, c, h, wd = x.shape
    if tuple(w.shape) != (c, 9) or (bias is not None and bias.numel() != c):
        raise RuntimeError(f"dwconv3x3: weight {tuple(w.shape)} / bias do not match {c} channels")
    if bias is not None:
        _req(bias, "bias")
    oh, ow = (h + stride - 1) // stride, (wd + stride - 1) // stride
    y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_dwconv3x3_fwd(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), n, c, h, wd, int(stride), {"none": 0, "silu": 3}[act],
                                        _stream()), "slu_dwconv3x3_fwd")
    return y


def dwconv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """dL/dw [C, 9] of the stride-1 depthwise 3x3 conv y = dwconv3x3(x, w)."""
    _req(x, "x")
    _req(dy, "dy")
    if x.shape != dy.shape:
        raise RuntimeError("dwconv3x3_wgrad: x / dy shape mismatch")
    n, c, h, w = x.shape
    dw = torch.empty((c, 9), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_dwconv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, c, h, w, _stream()), "slu_dwconv3x3_wgrad")
    return dw


def global_avgpool(x: torch.Tensor) -> torch.Tensor:
    """mean over H W -> [N, C]."""
    _req(x, "x")
    n, c, h, w = x.shape
    avg = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(_lib.load().slu_global_avgpool(x.data_ptr(), avg.data_ptr(), n, c, h * w, _stream()), "slu_global_avgpool")
    return avg


def se_scale(x: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor]) -> torch.Tensor:
    """SqueezeExcitation's multiplier [N, C] = sigmoid(fc2(SiLU(fc1(mean_HW(x))))); w1 [S, C], w2 [C, S]."""
    _req(x, "x")
    n, c, h, w = x.shape
    s = w1.shape[0]
    for t, nme, shp in ((w1, "w1", (s, c)), (w2, "w2", (c, s))):
        _req(t, nme)
        if tuple(t.shape) != shp:
            raise RuntimeError(f"se_scale: {nme} expected {shp}, got {tuple(t.shape)}")
    lib = _lib.load()
    avg = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(lib.slu_global_avgpool(x.data_ptr(), avg.data_ptr(), n, c, h * w, _stream()), "slu_global_avgpool")
    scale = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(lib.slu_se_gate(avg.data_ptr(), w1.data_ptr(), _ptr(b1), w2.data_ptr(), _ptr(b2), scale.data_ptr(), n, c, s, _stream()), "slu_se_gate")
    return scale
