"""Device ops of the accelerated path: thin wrappers over the libfsc_hip.so C ABI plus the
``torch.autograd.Function``s that give them gradients.  PyTorch is plumbing here (device
memory, streams, the autograd tape); every number is produced by a hand-written HIP kernel.

Layout: activations are NCHW fp32 like the reference; the 1-d model runs with H == 1.
"""
import ctypes as C
import os
import sys
import threading
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import ConvDesc, call, ptr, stream_ptr

BN_EPS = 1e-5
LOG_EPS = 1e-4


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def _need_cuda(t, what):
    if not t.is_cuda:
        raise _lib.FscError("%s: expected a device tensor, got %s (no CPU fallback)" % (what, t.device))


# ------------------------------------------------------------------------------ front-end
_TABLES = {}


def _frontend_tables(n_fft, device):
    key = (n_fft, str(device))
    if key not in _TABLES:
        n = _lib.load().fsc_frontend_table_floats(n_fft)
        t = torch.empty(n, device=device, dtype=torch.float32)
        call("fsc_frontend_tables_init", ptr(t), n_fft, stream_ptr())
        _TABLES[key] = t
    return _TABLES[key]


class MelBands:
    """Banded view of a dense (n_mel, n_bins) filterbank: per row the first non-zero bin, the
    band length, and the weights stored [j][m] so lanes that own consecutive mel rows read
    consecutive addresses.  Host-side index work only."""

    def __init__(self, filterbank, device):
        fb = np.asarray(filterbank, dtype=np.float32)
        n_mel = fb.shape[0]
        start = np.zeros(n_mel, np.int32)
        length = np.zeros(n_mel, np.int32)
        for m in range(n_mel):
            nz = np.flatnonzero(fb[m])
            if nz.size:
                start[m] = nz[0]
                length[m] = nz[-1] - nz[0] + 1
        max_band = int(length.max()) if n_mel else 0
        w = np.zeros((max(max_band, 1), n_mel), np.float32)
        for m in range(n_mel):
            w[:length[m], m] = fb[m, start[m]:start[m] + length[m]]
        self.n_mel = n_mel
        self.n_bins = fb.shape[1]
        self.max_band = max_band
        self.start = torch.from_numpy(start).to(device)
        self.length = torch.from_numpy(length).to(device)
        self.weights = torch.from_numpy(w).to(device)


def frontend_logmel(wave, n_fft, hop, bands, freq_channel):
    """(N, T) waveform -> (N, 1|2, n_mel, frames) log-mel (+ frequency-encoding channel)."""
    _need_cuda(wave, "frontend_logmel")
    wave = wave.contiguous()
    n, t = wave.shape
    frames = 1 + t // hop
    planes = 2 if freq_channel else 1
    out = _empty((n, planes, bands.n_mel, frames), wave)
    tables = _frontend_tables(n_fft, wave.device)
    with _stage("frontend", 4 * n * t + 4 * n * bands.n_mel * frames):      # (the frequency channel is synthesised)
        call("fsc_frontend_logmel_fwd", ptr(wave), n, t, t, n_fft, hop,
             ptr(tables), ptr(bands.start), ptr(bands.length),
             ptr(bands.weights), bands.n_mel, bands.max_band, LOG_EPS, ptr(out),
             planes * bands.n_mel * frames, 1 if freq_channel else 0, stream_ptr())
    return out


def frontend_stft(wave, n_fft, hop, apply_log, freq_channel=False):
    """(N, T) waveform -> (N, [1|2,] n_fft/2+1, frames) magnitude or log-magnitude."""
    _need_cuda(wave, "frontend_stft")
    wave = wave.contiguous()
    n, t = wave.shape
    frames = 1 + t // hop
    bins = n_fft // 2 + 1
    if freq_channel:
        out = _empty((n, 2, bins, frames), wave)
        stride = 2 * bins * frames
    else:
        out = _empty((n, bins, frames), wave)
        stride = bins * frames
    tables = _frontend_tables(n_fft, wave.device)
    with _stage("frontend", 4 * n * t + 4 * n * bins * frames):
        call("fsc_frontend_stft_fwd", ptr(wave), n, t, t, n_fft, hop,
             ptr(tables), 1 if apply_log else 0, LOG_EPS, ptr(out),
             stride, 1 if freq_channel else 0, stream_ptr())
    return out


# ------------------------------------------------------------------------------ convolution
_ARITH = None        # host-side default for the descriptors built here (None: the library's default -- FSC_CONV_ARITH, else f16x6)


def _desc(n, c_in, c_out, h, w, kh, kw, arith=None):
    """fsc_conv_desc; the arithmetic mode is a per-call field of the descriptor (no library state)."""
    return ConvDesc(n, c_in, c_out, h, w, kh, kw, get_conv_arith() if arith is None else arith)


class KernelTimer:
    """Optional per-launch timing of the conv kernels with HIP events recorded on the stream
    the kernels are launched on (bench.py uses it for the roofline numbers).  Off by default."""

    def __init__(self):
        self.records = []      # (kernel name, algorithmic flops, start event, end event)
        self.shapes = {}       # kernel name -> (flops, (n, c_in, c_out, h, w, kh, kw), "fwd" | "dgrad" | "wgrad") of its largest call

        self.bytes = {}        # kernel name -> [algorithmic HBM bytes of its noted calls, noted calls]

    def note(self, name, flops, shape, kind):
        if name not in self.shapes or flops > self.shapes[name][0]:
            self.shapes[name] = (flops, tuple(shape), kind)
        # algorithmic bytes of the call: each operand read once, the result written once (three-limb operands 6 B, others 4 B
        # per element; the fused max-pool writes a quarter of the plane plus one index byte)
        n, c_in, c_out, h, w, kh, kw = shape
        b_in = 6.0 if "conv_l3_" in name else 4.0
        px = float(n) * h * w
        if kind == "wgrad":
            nb = px * (c_in + c_out) * b_in + 4.0 * c_out * c_in * kh * kw
        elif kind == "dgrad":
            nb = px * c_out * b_in + px * c_in * 4.0
        else:
            nb = px * c_in * b_in + (px / 4.0 * c_out * 5.0 if "pool" in name else px * c_out * 4.0)
        acc = self.bytes.setdefault(name, [0.0, 0])
        acc[0] += nb
        acc[1] += 1

    def summary(self):
        """name -> dict(launches, flops, ms); call after a device synchronise."""
        out = {}
        for name, flops, e0, e1 in self.records:
            r = out.setdefault(name, dict(launches=0, flops=0.0, ms=0.0))
            r["launches"] += 1
            r["flops"] += flops
            r["ms"] += e0.elapsed_time(e1)
        return out


TIMER = None


class StageTimer:
    """Optional timing of the HBM-bound stages (front-end, BatchNorm / PReLU / pooling passes, optimizer) with HIP events on the
    launch stream, next to the ALGORITHMIC bytes of each call: every tensor a pass reads or writes counted once (a backward call
    is two passes: reduce + apply).  bench.py runs a few extra steps under it, outside the timed region, for the per-stage
    `roofline.stages` block.  Off by default."""

    def __init__(self):
        self.records = []      # (stage, bytes, start event, end event)

    def summary(self):
        out = {}
        for name, nbytes, e0, e1 in self.records:
            r = out.setdefault(name, dict(calls=0, bytes=0.0, ms=0.0))
            r["calls"] += 1
            r["bytes"] += nbytes
            r["ms"] += e0.elapsed_time(e1)
        return out


STAGE_TIMER = None


class _stage:
    def __init__(self, name, nbytes):
        self.on = STAGE_TIMER is not None
        self.name, self.nbytes = name, nbytes

    def __enter__(self):
        if self.on:
            # the stream is drained first: an event behind a long kernel (a convolution) is stamped while that kernel still runs
            # and the pair would charge its tail to this stage (measured: 390 us for a 5 us finalisation).  Stage timing is a
            # separate diagnostic pass, never part of a timed region.
            torch.cuda.current_stream().synchronize()
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            nbytes = float(self.nbytes() if callable(self.nbytes) else self.nbytes)
            if nbytes > 0:
                STAGE_TIMER.records.append((self.name, nbytes, self.e0, self.e1))


def _nb(*tensors):
    """Bytes of the given tensors (None skipped; L16 tensors by their data buffer)."""
    total = 0
    for t in tensors:
        if t is None:
            continue
        t = t.data if isinstance(t, L16) else t
        total += t.numel() * t.element_size()
    return total


def plan_name(desc, mode):
    buf = C.create_string_buffer(256)
    call("fsc_conv_plan_describe", C.byref(desc), mode, buf, 256)
    return buf.value.decode().split(" ")[0]


class _timed:
    def __init__(self, desc, mode):
        self.on = TIMER is not None
        if self.on:
            self.name = plan_name(desc, mode)
            self.flops = 2.0 * desc.n * desc.h * desc.w * desc.c_in * desc.c_out * desc.kh * desc.kw
            # (algorithmic bytes of the call: fp32 operands once + the result once)
            TIMER.note(self.name, self.flops, (desc.n, desc.c_in, desc.c_out, desc.h, desc.w, desc.kh, desc.kw),
                       ("fwd", "dgrad", "wgrad")[mode] if mode in (0, 1, 2) else "fwd")

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            TIMER.records.append((self.name, self.flops, self.e0, self.e1))


ARITH_NAMES = {"f32": 0, "bf16": 1, "f16x3": 3, "bf16x6": 6, "bf16x9": 9, "f16x6": 10}


def set_conv_arith(mode):
    """Arithmetic the descriptors built by this module ask for (fsc_conv_desc.arith, include/fsc_hip.h).
    0 / "f32": native fp32 MFMA; 3 / "f16x3" (the opt-in FAST mode: 22-bit products): fp32 via two fp16 limbs with exact
    power-of-two operand scaling, three limb products; 6, 9 / "bf16x6", "bf16x9": fp32 via exact three-limb bf16 split with
    that many limb products; 10 / "f16x6": three SCALED fp16 limbs, six products, on the pre-split (L16) route -- fp32-equivalent
    (products to 2^-32) at two thirds of bf16x9's matrix work, bf16x9 wherever a layer has no L16 kernel -- THE LIBRARY DEFAULT
    (round 6: the arithmetic bench.py's `value` is measured in is the one that ships); 1 / "bf16": plain bf16 operands, fp32
    accumulation (mixed precision, cfg 3); None: back to the library default (FSC_CONV_ARITH or 10).  Host-side convenience only:
    the C ABI takes the mode per call."""
    global _ARITH
    mode = ARITH_NAMES.get(mode, mode)
    if mode is not None and int(mode) not in (0, 1, 3, 6, 9, 10):
        raise _lib.FscError("conv arithmetic must be 0 (f32), 1 (bf16), 3 (f16x3), 6 (bf16x6), 9 (bf16x9) or 10 (f16x6); got %r" % (mode,))
    _ARITH = None if mode is None else int(mode)


def get_conv_arith():
    return _lib.load().fsc_conv_default_arith() if _ARITH is None else _ARITH


AMAX_FLOATS = 512          # include/fsc_hip.h FSC_AMAX_FLOATS


def amax(x):
    """max |x| as an FSC_AMAX_FLOATS slot buffer (fsc_amax; the value is the maximum over the buffer)."""
    out = _empty((AMAX_FLOATS,), x)
    call("fsc_amax", ptr(x), x.numel(), ptr(out), stream_ptr())
    return out


def _operand_amax(x, given):
    """The split-fp16 conv kernels scale their operands by the tensor's largest magnitude; producers that
    know it pass it along, otherwise it costs one read of the tensor."""
    if given is not None or get_conv_arith() != 3:
        return given
    return amax(x)


def _desc_key(desc):
    return tuple(getattr(desc, f) for f, _ in desc._fields_)


def conv_pack(desc, weight, dgrad):
    hit = _XPACKED.pop((weight.data_ptr(), dgrad), None)
    if hit is not None and hit[0] == _desc_key(desc) and hit[2] == weight._version:
        return hit[1]                                   # packed up front with the step's other weights (prepack_begin)
    lib = _lib.load()
    n = lib.fsc_conv_packed_floats(C.byref(desc), dgrad)
    if n == 0:
        raise _lib.FscError("conv: unsupported shape %s" % [getattr(desc, f) for f, _ in desc._fields_])
    if _XPACK_RECORD is not None and lib.fsc_conv_pack_weights_multi_supported(C.byref(desc), dgrad):
        _XPACK_RECORD.append((weight, _desc_key(desc), dgrad))
    packed = _empty((n,), weight)
    call("fsc_conv_pack_weights", C.byref(desc), ptr(weight), dgrad, ptr(packed), stream_ptr())
    return packed


# Training on the bf16-limb kernels (cfg 3): every convolution packs its weight per use, forward and input gradient -- 80 launches
# of ~4.6 us per step of the 1-d model.  Like the L16 fragments below, the first training forward AND backward of a (model, input
# shape) record which (weight, descriptor, direction) they pack; from then on prepack_begin() packs them all with
# fsc_conv_pack_weights_multi (two launches) and conv_pack() hands the fragments out -- the input-gradient ones in the backward of
# the same step (the weights do not move in between; an entry whose weight has a newer version is dropped).
_XPACK_PLAN = {}       # key -> [(weight, desc key, dgrad), ...]
_XPACK_RECORD = None
_XPACKED = {}          # (weight.data_ptr(), dgrad) -> (desc key, packed, weight version)


def _xpack_begin(key):
    """The x-path half of prepack_begin: True when this step records."""
    global _XPACK_RECORD
    _XPACKED.clear()
    _XPACK_RECORD = None
    if get_conv_arith() != 1:
        return False
    plan = _XPACK_PLAN.get(key)
    if plan is None:
        _XPACK_RECORD = []
        _XPACK_PLAN[key] = _XPACK_RECORD               # (filled by this step's forward and backward)
        if len(_XPACK_PLAN) > 64:
            for k in [k for k in _XPACK_PLAN if k != key]:
                del _XPACK_PLAN[k]
        return True
    if not plan:
        return False
    count = len(plan)
    descs = (ConvDesc * count)()
    wp, pp, dg = (C.c_void_p * count)(), (C.c_void_p * count)(), (C.c_int * count)()
    lib = _lib.load()
    made = []
    for i, (weight, dkey, dgrad) in enumerate(plan):
        d = ConvDesc(*dkey)
        packed = _empty((lib.fsc_conv_packed_floats(C.byref(d), dgrad),), weight)
        descs[i] = d
        wp[i], pp[i], dg[i] = ptr(weight), ptr(packed), dgrad
        made.append((weight, dkey, dgrad, packed))
    call("fsc_conv_pack_weights_multi", count, descs, wp, dg, pp, stream_ptr())
    for weight, dkey, dgrad, packed in made:
        _XPACKED[(weight.data_ptr(), dgrad)] = (dkey, packed, weight._version)
    return False


_X3_STATS_LAYOUT = {}


def conv_forward(x, weight, bias, x_amax=None, stats_bn=None):
    """x (N, Cin, H, W), weight (Cout, Cin, kh, kw) -> (N, Cout, H, W); stride 1, same pad.
    stats_bn = (BatchNorm module, training): where the layer has such a kernel (plain bf16 on 1-d rows, ring-kernel tiling:
    fsc_conv_fwd_stats) the epilogue also reduces that BatchNorm's statistics of the output, picked up by bn_prepare on this
    very tensor."""
    n, c_in, h, w = x.shape
    c_out, _, kh, kw = weight.shape
    d = _desc(n, c_in, c_out, h, w, kh, kw)
    packed = conv_pack(d, weight, 0)
    out = _empty((n, c_out, h, w), x)
    if stats_bn is not None and CONV_STATS and d.arith == 1 and h == 1 and stats_bn[1]:
        key = _desc_key(d)
        if key not in _X3_STATS_LAYOUT:
            out4 = (C.c_int * 4)()
            _X3_STATS_LAYOUT[key] = tuple(out4) if _lib.load().fsc_conv_fwd_stats_layout(C.byref(d), out4) else None
        lay = _X3_STATS_LAYOUT[key]
        if lay is not None:
            bn, training = stats_bn
            rec = torch.empty(lay[0] * 8 * lay[1] * lay[2] * 4, device=x.device, dtype=torch.float32)     # ([channel][slot])
            pivot = bn.running_mean if bn_tracks(bn, training) else None
            with _timed(d, 0):
                call("fsc_conv_fwd_stats", C.byref(d), ptr(x), ptr(packed), ptr(bias), ptr(out), ptr(pivot), ptr(rec), stream_ptr())
            _stats_end(lay, rec, out, c_out)
            return out
    x_amax = _operand_amax(x, x_amax)
    with _timed(d, 0):
        call("fsc_conv_fwd", C.byref(d), ptr(x), ptr(packed), ptr(bias), 0, 0, ptr(out), ptr(x_amax), stream_ptr())
    return out


def conv_pool_forward(x, weight, bias):
    """Stem conv 3x3 fused with MaxPool2d(2) (classifiers.py:526-532): (pooled, window index, conv output
    shape), or None when the shape is not a stem layer (the caller then runs conv_forward + maxpool_forward)."""
    n, c_in, h, w = x.shape
    c_out, _, kh, kw = weight.shape
    d = _desc(n, c_in, c_out, h, w, kh, kw)
    if not _lib.load().fsc_conv_pool_supported(C.byref(d)):
        return None
    packed = conv_pack(d, weight, 0)
    y = _empty((n, c_out, h // 2, w // 2), x)
    idx = _empty((n, c_out, h // 2, w // 2), x, torch.uint8)
    with _timed(d, 0):
        call("fsc_conv_pool_fwd", C.byref(d), ptr(x), ptr(packed), ptr(bias), ptr(y), ptr(idx), stream_ptr())
    return y, idx, (n, c_out, h, w)


CONV_POOL_1D = os.environ.get("FSC_CONV_POOL_1D", "1") != "0"
_POOL1D_OK = {}


def conv_pool1d_forward(x, weight, bias, stats_bn):
    """k3 Conv1d fused with MaxPool1d(2) and the statistics of the BatchNorm behind the pool (classifiers.py:149-156; plain bf16,
    ring-kernel tiling: fsc_conv_fwd_pool_stats): (pooled, window index, conv output shape) with the records left for bn_prepare on
    the pooled tensor, or None (the caller then runs the convolution, the pool and the statistics pass)."""
    n, c_in, h, w = x.shape
    c_out, _, kh, kw = weight.shape
    if not (CONV_POOL_1D and CONV_STATS and h == 1 and kh == 1 and kw == 3 and stats_bn is not None and stats_bn[1]):
        return None
    d = _desc(n, c_in, c_out, h, w, kh, kw)
    if d.arith != 1:
        return None
    key = _desc_key(d)
    if key not in _POOL1D_OK:
        out4 = (C.c_int * 4)()
        ok = _lib.load().fsc_conv_fwd_pool_stats_supported(C.byref(d)) and _lib.load().fsc_conv_fwd_stats_layout(C.byref(d), out4)
        _POOL1D_OK[key] = tuple(out4) if ok else None
    lay = _POOL1D_OK[key]
    if lay is None:
        return None
    bn, training = stats_bn
    packed = conv_pack(d, weight, 0)
    y = _empty((n, c_out, 1, w // 2), x)
    idx = _empty((n, c_out, 1, w // 2), x, torch.uint8)
    rec = torch.empty(lay[0] * 8 * lay[1] * lay[2] * 4, device=x.device, dtype=torch.float32)
    pivot = bn.running_mean if bn_tracks(bn, training) else None
    with _timed(d, 0):
        call("fsc_conv_fwd_pool_stats", C.byref(d), ptr(x), ptr(packed), ptr(bias), ptr(y), ptr(idx), ptr(pivot), ptr(rec), stream_ptr())
    _stats_end(lay, rec, y, c_out)
    return y, idx, (n, c_out, h, w)


def conv_dgrad(dout, weight, x_shape, accumulate_into=None, dout_amax=None):
    """Gradient w.r.t. the conv input.  With `accumulate_into` the result is added in place."""
    n, c_in, h, w = x_shape
    c_out, _, kh, kw = weight.shape
    d = _desc(n, c_in, c_out, h, w, kh, kw)
    packed = conv_pack(d, weight, 1)
    if accumulate_into is None:
        dx = _empty((n, c_in, h, w), dout)
        acc = 0
    else:
        dx = accumulate_into
        acc = 1
    dout_amax = _operand_amax(dout, dout_amax)
    with _timed(d, 1):
        call("fsc_conv_fwd", C.byref(d), ptr(dout), ptr(packed), None, 1, acc, ptr(dx), ptr(dout_amax), stream_ptr())
    return dx


# ---- pre-split activations (include/fsc_hip.h "L16" tensors)
class L16:
    """An activation held as 16-bit limbs in the MFMA operand layout: two scaled fp16 limbs plus the maximum the scale derives
    from (arith 3; limbs = 2), three exact bf16 limbs (arith 9: no scale, amax is None; limbs = 3) or three scaled fp16 limbs
    (arith 10; limbs = 4, the format code FSC_L16_F16X3 of include/fsc_hip.h)."""
    __slots__ = ("data", "amax", "shape", "limbs")

    def __init__(self, data, amax, shape, limbs=2):
        self.data, self.amax, self.shape, self.limbs = data, amax, tuple(shape), limbs


def _l16_arith():
    """The arithmetic of the pre-split (L16) route under the current mode: 3 (two scaled fp16 limbs, three products), 9 (three
    exact bf16 limbs, all nine products: fp32-exact), 10 (three scaled fp16 limbs, six products), or None when the mode has no
    L16 kernels."""
    a = get_conv_arith()
    return a if a in (3, 9, 10) else None


def _l16_limbs(arith=None):
    """The L16 format code (the `limbs` argument of include/fsc_hip.h) of an arithmetic: 2, 3, or 4 (three scaled fp16 limbs)."""
    return {9: 3, 10: 4}.get(_l16_arith() if arith is None else arith, 2)


BN_L16_LIMBS3 = 64        # include/fsc_hip.h FSC_BN_L16_LIMBS3
BN_L16_F16X3 = 128        # include/fsc_hip.h FSC_BN_L16_F16X3
_BN_L16_FLAG = {3: BN_L16_LIMBS3, 4: BN_L16_F16X3}


def l16_empty(shape, like, limbs=None):
    n, c = shape[0], shape[1]
    hw = 1
    for v in shape[2:]:
        hw *= v
    nbytes = _lib.load().fsc_l16_bytes_limbs(n, c, hw, _l16_limbs() if limbs is None else limbs)
    return torch.empty(nbytes // 4, device=like.device, dtype=torch.int32)


def l16_pack(x, x_amax=None, limbs=None):
    """fp32 (N, C, ...) -> L16 (one read + one write; the fused producers write the format directly)."""
    limbs = _l16_limbs() if limbs is None else limbs
    if limbs != 3:
        x_amax = amax(x) if x_amax is None else x_amax
    else:
        x_amax = None
    n, c = x.shape[0], x.shape[1]
    data = l16_empty(x.shape, x, limbs)
    call("fsc_l16_pack_limbs", ptr(x), n, c, x.numel() // (n * c), ptr(x_amax), limbs, ptr(data), stream_ptr())
    return L16(data, x_amax, x.shape, limbs)


def l16_unpack(t):
    n, c = t.shape[0], t.shape[1]
    x = torch.empty(t.shape, device=t.data.device, dtype=torch.float32)
    call("fsc_l16_unpack_limbs", ptr(t.data), n, c, x.numel() // (n * c), ptr(t.amax), t.limbs, ptr(x), stream_ptr())
    return x


def conv_l16_supported(desc, dgrad):
    return bool(_lib.load().fsc_conv_l16_supported(C.byref(desc), dgrad))


def l16_plan_name(desc, dgrad):
    buf = C.create_string_buffer(256)
    call("fsc_conv_l16_plan_describe", C.byref(desc), dgrad, buf, 256)
    return buf.value.decode().split(" ")[0]


# Inference: the forward fragments of a weight are reused across batches.  The fragment layout depends on the tiling only through
# (channel tiles per block, channel blocks), so length-grouped batches of different shapes share one packing.  Entries are keyed by
# the weight's address AND version; the library's own optimizers write through raw pointers (no version bump) and therefore call
# forget_packed_weights() in their step().
_EVAL_PACKS = {}
_PLAN_SIG = {}


_EVAL_BN = {}


def forget_packed_weights():
    """THE invalidation hook of the inference caches (packed weight fragments, folded eval-mode BatchNorm scale / shift, the
    calibrated operand ranges of the folded conv -> BatchNorm -> PReLU launches).  The
    caches are keyed by address and torch version counter, so every in-place torch op invalidates them by itself; anything that
    writes parameters or BatchNorm buffers through raw pointers (this library's optimizers do, an external EMA / SWA kernel would)
    must call this.  Called by the Fused* optimizers' step(), a training-mode bn_prepare, and the model's train() /
    load_state_dict() / load_best_model() / close()."""
    global _XPACK_RECORD
    _EVAL_PACKS.clear()
    _EVAL_BN.clear()
    _ACT_CAL.clear()          # (calibrated output ranges of the folded inference launches: they depend on every parameter in front)
    _XPACKED.clear()          # (fragments packed up front for a step are that step's)
    _XPACK_RECORD = None      # (and the recording of a first step ends with its optimizer step)


def _l16_plan_sig(d):
    key = tuple(getattr(d, f) for f, _ in d._fields_)
    if key not in _PLAN_SIG:
        buf = C.create_string_buffer(256)
        call("fsc_conv_l16_plan_describe", C.byref(d), 0, buf, 256)
        txt = buf.value.decode()                     # conv_l16_fwd_kernel<kh,kw,cot,pt> box=... items=TILESxBLOCKS ...
        cot = int(txt.split("<")[1].split(">")[0].split(",")[2])
        if " cot=" in txt:                           # conv_l3_fwd_kernel<kh,kw,ct,ptw,nprod> cot=... : tiles per block spelled out
            cot = int(txt.split(" cot=")[1].split(" ")[0])
        blocks = int(txt.split("items=")[1].split(" ")[0].split("x")[1])
        _PLAN_SIG[key] = (cot, blocks)
    return _PLAN_SIG[key]


def conv_l16_pack(weight, n, h, w, dgrad):
    """Packed A fragments of `weight` for fsc_conv_l16_fwd on (n, ., h, w) activations: (descriptor, packed)."""
    c_out, c_in, kh, kw = weight.shape
    d = _desc(n, c_in, c_out, h, w, kh, kw, _l16_arith())
    dg = 1 if dgrad else 0
    nfl = _lib.load().fsc_conv_l16_packed_floats(C.byref(d), dg)
    if nfl == 0:
        raise _lib.FscError("conv_l16: unsupported shape %s" % [getattr(d, f) for f, _ in d._fields_])
    key = None
    if not dgrad and not torch.is_grad_enabled():
        key = (weight.data_ptr(), weight._version, tuple(weight.shape), nfl, _l16_arith()) + _l16_plan_sig(d)
        hit = _EVAL_PACKS.get(key)
        if hit is not None:
            return d, hit[0]
    packed = torch.empty(nfl, device=weight.device, dtype=torch.float32)
    call("fsc_conv_l16_pack_weights", C.byref(d), ptr(weight), dg, ptr(packed), stream_ptr())
    if key is not None:
        if len(_EVAL_PACKS) > 4096:
            _EVAL_PACKS.clear()
        _EVAL_PACKS[key] = (packed, weight)          # (the weight is kept alive: its address cannot be reused meanwhile)
    return d, packed


# Training: the weights of every L16 convolution are re-packed each step.  The first forward of a (model, input shape) records
# which (weight, n, h, w) it packs; from then on prepack_begin() packs them all up front with fsc_conv_l16_pack_weights_multi
# (6 launches instead of 40 at cfg 2) and conv_l16_pack_pair() hands the fragments out.
MULTI_PACK = os.environ.get("FSC_MULTI_PACK", "1") == "1"
_PACK_PLAN = {}        # key -> [(weight, n, h, w), ...]
_PACK_RECORD = None    # list being recorded, or None
_PREPACKED = {}        # weight.data_ptr() -> ((n, h, w), fwd or None, dgrad or None), consumed once


def prepack_begin(key):
    """Call at the start of a training forward; `key` identifies (model, input shape).  Returns a token for prepack_end."""
    global _PACK_RECORD
    _PREPACKED.clear()
    if MULTI_PACK:
        _xpack_begin(key)
    if not MULTI_PACK or _l16_arith() is None or not USE_L16:
        return None
    plan = _PACK_PLAN.get(key)
    if plan is None:
        _PACK_RECORD = []
        return ("record", key)
    lib = _lib.load()
    count = len(plan)
    descs = (ConvDesc * count)()
    wp, fp, dp = (C.c_void_p * count)(), (C.c_void_p * count)(), (C.c_void_p * count)()
    made = []
    for i, (weight, n, h, w) in enumerate(plan):
        c_out, c_in, kh, kw = weight.shape
        d = _desc(n, c_in, c_out, h, w, kh, kw, _l16_arith())
        nf, nd = lib.fsc_conv_l16_packed_floats(C.byref(d), 0), lib.fsc_conv_l16_packed_floats(C.byref(d), 1)
        pf = torch.empty(nf, device=weight.device, dtype=torch.float32) if nf else None
        pd = torch.empty(nd, device=weight.device, dtype=torch.float32) if nd else None
        descs[i] = d
        wp[i], fp[i], dp[i] = ptr(weight), ptr(pf), ptr(pd)
        made.append((weight, (n, h, w), d, pf, pd))
    call("fsc_conv_l16_pack_weights_multi", count, descs, wp, fp, dp, stream_ptr())
    for weight, shape, d, pf, pd in made:
        _PREPACKED[weight.data_ptr()] = (shape, (d, pf) if pf is not None else None, (d, pd) if pd is not None else None)
    return ("packed", key)


def prepack_end(token):
    global _PACK_RECORD
    if token is not None and token[0] == "record" and _PACK_RECORD is not None:
        if _PACK_RECORD:
            if len(_PACK_PLAN) > 64:
                _PACK_PLAN.clear()
            _PACK_PLAN[token[1]] = _PACK_RECORD
    if token is not None and token[0] == "packed" and _PREPACKED:
        _PACK_PLAN.pop(token[1], None)       # fragments nobody asked for: the plan is stale (another model at this id) -- re-record
    _PACK_RECORD = None
    _PREPACKED.clear()


def prepack_forget(model_id):
    global _XPACK_RECORD
    for key in [k for k in _PACK_PLAN if k[0] == model_id]:
        del _PACK_PLAN[key]
    for key in [k for k in _XPACK_PLAN if k[0] == model_id]:
        del _XPACK_PLAN[key]
    _XPACK_RECORD = None
    _XPACKED.clear()


def conv_l16_pack_pair(weight, n, h, w):
    """Forward and input-gradient fragments of one weight in one call: ((desc, packed_fwd) or None, (desc, packed_dgrad) or
    None), for the directions fsc_conv_l16_fwd has a tiling for."""
    hit = _PREPACKED.pop(weight.data_ptr(), None)
    if hit is not None and hit[0] == (n, h, w):
        return hit[1], hit[2]
    c_out, c_in, kh, kw = weight.shape
    d = _desc(n, c_in, c_out, h, w, kh, kw, _l16_arith())
    lib = _lib.load()
    nf, nd = lib.fsc_conv_l16_packed_floats(C.byref(d), 0), lib.fsc_conv_l16_packed_floats(C.byref(d), 1)
    if nf == 0 and nd == 0:
        return None, None
    if _PACK_RECORD is not None:
        _PACK_RECORD.append((weight, n, h, w))
    pf = torch.empty(nf, device=weight.device, dtype=torch.float32) if nf else None
    pd = torch.empty(nd, device=weight.device, dtype=torch.float32) if nd else None
    call("fsc_conv_l16_pack_weights_pair", C.byref(d), ptr(weight), ptr(pf), ptr(pd), stream_ptr())
    return ((d, pf) if nf else None), ((d, pd) if nd else None)


EVAL_L16 = os.environ.get("FSC_EVAL_L16", "1") == "1"          # inference also takes the L16 kernels (needs the range of each BN input)
_STATS_MINMAX_ONLY = 16    # FSC_BN_STATS_MINMAX_ONLY
CONV_STATS = os.environ.get("FSC_CONV_STATS", "1") == "1"      # forward convolutions reduce the statistics of the BatchNorm they feed
_STATS_PIVOT_RM = 8        # FSC_BN_STATS_PIVOT_RM
_STATS_LAYOUT = {}


def _stats_layout(d, pool):
    """(workers, channel blocks, channels per block) of the statistics records of a STATS convolution, or None."""
    key = tuple(getattr(d, f) for f, _ in d._fields_) + (pool,)
    if key not in _STATS_LAYOUT:
        out3 = (C.c_int * 4)()
        ok = _lib.load().fsc_conv_l16_stats_layout(C.byref(d), 1 if pool else 0, out3)
        _STATS_LAYOUT[key] = tuple(out3) if ok else None
    return _STATS_LAYOUT[key]


def bn_tracks(bn, training):
    return bool(training and bn.track_running_stats and bn.running_mean is not None)


def _stats_begin(d, pool, stats_bn, like):
    """Records buffer and pivot of a STATS convolution feeding `stats_bn` = (BatchNorm, training), or (None, None, None)."""
    if not CONV_STATS or stats_bn is None:
        return None, None, None
    bn, training = stats_bn
    if not (training or bn.running_mean is None or EVAL_L16):     # (eval: only the range of the output is of use, for L16)
        return None, None, None
    lay = _stats_layout(d, pool)
    if lay is None:
        return None, None, None
    rec = torch.empty(lay[0] * 8 * lay[2] * 4, device=like.device, dtype=torch.float32)
    return lay, rec, (bn.running_mean if bn_tracks(bn, training) else None)


_STATS_CONV_REC = 32       # (host-side flag) the entry holds the raw records of a STATS convolution, not a folded workspace


def _stats_end(lay, rec, y, c_out):
    """Leaves the records of the convolution that wrote `y` for bn_prepare(y, ...): folded and finalised there in one launch
    (fsc_bn_train_stats_conv), or folded into a BatchNorm workspace first where that entry point does not apply."""
    _PRESTATS.clear()
    _PRESTATS[(y.data_ptr(), tuple(y.shape), y._version, y.device)] = ((rec, lay, c_out), y, _STATS_CONV_REC | _STATS_FOLDED | _STATS_PIVOT_RM)


def _fold_conv_records(entry, like):
    rec, lay, c_out = entry
    ws = _bn_ws(c_out, like)
    call("fsc_bn_records_fold_conv", ptr(rec), lay[0], lay[1], lay[2], lay[3], c_out, ptr(ws), stream_ptr())
    return ws


def conv_l16(t, weight, bias, dgrad=False, accumulate_into=None, prepacked=None, stats_bn=None):
    """Forward (dgrad False: t = input, (N, Cin, H, W)) or input gradient (dgrad True: t = dout (N, Cout, H, W)) of a
    stride-1 same-pad convolution on an L16 operand; fp32 NCHW result.  stats_bn = (BatchNorm module, training) (forward only):
    the kernel also reduces that BatchNorm's statistics of the output (picked up by bn_prepare on this very tensor)."""
    c_out, c_in, kh, kw = weight.shape
    n, _, h, w = t.shape
    dg = 1 if dgrad else 0
    d, packed = prepacked if prepacked is not None else conv_l16_pack(weight, n, h, w, dgrad)
    if accumulate_into is not None:
        out, acc = accumulate_into, 1
    else:
        out, acc = torch.empty((n, c_in if dgrad else c_out, h, w), device=weight.device, dtype=torch.float32), 0
    lay, rec, pivot = _stats_begin(d, False, stats_bn, out) if (not dgrad and acc == 0) else (None, None, None)
    if TIMER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if lay is not None:
        call("fsc_conv_l16_fwd_stats", C.byref(d), ptr(t.data), ptr(t.amax), ptr(packed), ptr(bias), ptr(out), ptr(pivot), ptr(rec),
             stream_ptr())
    else:
        call("fsc_conv_l16_fwd", C.byref(d), ptr(t.data), ptr(t.amax), ptr(packed), ptr(bias), dg, acc, ptr(out), stream_ptr())
    if TIMER is not None:
        e1.record()
        TIMER.records.append((l16_plan_name(d, dg), 2.0 * n * h * w * c_in * c_out * kh * kw, e0, e1))
        TIMER.note(l16_plan_name(d, dg), 2.0 * n * h * w * c_in * c_out * kh * kw, (n, c_in, c_out, h, w, kh, kw), "dgrad" if dgrad else "fwd")
    if lay is not None:
        _stats_end(lay, rec, out, c_out)
    return out


# ---- inference: convolution + eval-mode BatchNorm + PReLU written straight as the next convolution's L16 operand
# (fsc_conv_l16_fwd_act).  bf16 limbs (arith 9) carry no scale: the fold is unconditional.  Scaled fp16
# limbs (arith 3, 10) need the DECLARED maximum of the output before it exists: a calibrated bound -- what the two-pass route derived
# for this layer on an earlier batch, doubled -- and a check that nothing outgrew it.  The check reads device memory, so it belongs
# to the caller's own synchronisation point: the fold with a calibrated scale only runs inside an `act_fold_scope()`, whose
# `ok()` the caller asks once the results are on their way to the host anyway (predict_2d_cnn.ensemble_batch / predict_folds);
# a scope that is not ok() has dropped the calibrations that overflowed and the caller recomputes the batch.
EVAL_ACT_FOLD = os.environ.get("FSC_EVAL_ACT_FOLD", "1") == "1"
_ACT_CAL = {}


class _ActFoldTLS(threading.local):
    def __init__(self):
        self.scope = None


_ACT_TLS = _ActFoldTLS()


class ActFoldScope:
    """Collects (largest |y| written, limit) of every calibrated fold launched while it is the current scope.  The flag words
    live in fixed chunks of SLOTS floats that are NEVER reallocated: a view handed to a launched kernel stays the word `ok()`
    reads, on whatever stream the kernel runs (a grown-and-copied buffer lost the maxima of kernels on other streams)."""
    SLOTS = 256

    def __init__(self):
        self.chunks = []
        self.limits = []
        self.keys = []
        self._prev = None
        self._depth = 0

    def __enter__(self):
        if self._depth == 0:                  # (re-entrant: eval_checked enters the scope and hands it to a callee that enters it too)
            self._prev = _ACT_TLS.scope
            _ACT_TLS.scope = self
        self._depth += 1
        return self

    def __exit__(self, *exc):
        self._depth -= 1
        if self._depth == 0:
            _ACT_TLS.scope = self._prev
            self._prev = None
        return False

    def prepare(self, like, slots=256):
        """Allocate flag words for `slots` folded launches now, on the current stream (callers that fold on several streams:
        before they fork, sized models x folded layers, so that no chunk is first touched on a side stream)."""
        while len(self.chunks) * self.SLOTS < slots:
            self.chunks.append(torch.zeros(self.SLOTS, device=like.device, dtype=torch.float32))

    def slot(self, like, limit, key):
        i = len(self.keys)
        if i >= len(self.chunks) * self.SLOTS:       # (a further chunk: zeroed on the stream of the launch that uses it first)
            self.chunks.append(torch.zeros(self.SLOTS, device=like.device, dtype=torch.float32))
        self.limits.append(limit)
        self.keys.append(key)
        return self.chunks[i // self.SLOTS][i % self.SLOTS:i % self.SLOTS + 1]

    def bad_mask(self):
        """Device float tensor, one entry per folded launch: 1.0 where the largest value written outgrew the calibrated limit
        (None when nothing was folded).  Callers append it to the result they copy to the host anyway -- ONE device-to-host
        copy and one synchronisation per batch -- and hand the host copy to ok()."""
        if not self.keys:
            return None
        n = len(self.keys)
        seen = self.chunks[0][:n] if n <= self.SLOTS else torch.cat(self.chunks)[:n]
        return (seen > torch.cat(self.limits)).to(torch.float32)

    def ok(self, bad_host=None):
        """True when no calibrated scale was outgrown (without `bad_host` -- the host copy of bad_mask() -- it synchronises:
        reads the flags).  Otherwise the calibrations concerned are dropped -- the next forward takes the two-pass route for
        those layers and calibrates afresh -- and the caller recomputes."""
        if not self.keys:
            return True
        bad = (self.bad_mask().cpu() if bad_host is None else bad_host) > 0
        if not bool(bad.any()):
            return True
        for i in torch.nonzero(bad).flatten().tolist():
            _ACT_CAL.pop(self.keys[i], None)
        return False


EVAL_RECOMPUTES = 0      # batches recomputed on the two-pass route because a calibrated scale was outgrown (diagnostic counter)


def eval_checked(fn):
    """Run `fn()` -- an inference forward returning a device tensor -- inside an act_fold_scope and return its result on the
    HOST, verified: the overflow flags of the calibrated folds ride on the same device-to-host copy; if a calibrated scale was
    outgrown the forward is repeated on the two-pass route (no scope).  The contract of the folded route in one place, for
    model.predict() / model.evaluate() (networks/classifiers.py) and predict_2d_cnn.ensemble_batch_checked."""
    scope = act_fold_scope()
    with scope:
        out = fn(scope)
    bad = scope.bad_mask()
    if bad is None:
        return out.cpu()
    host = torch.cat([out.reshape(-1).to(torch.float32), bad]).cpu()
    if scope.ok(host[out.numel():]):
        return host[:out.numel()].reshape(out.shape).to(out.dtype)
    global EVAL_RECOMPUTES
    EVAL_RECOMPUTES += 1
    return fn(None).cpu()


def act_fold_scope():
    return ActFoldScope()


def conv_l16_act_supported(t_shape, weight):
    if not EVAL_ACT_FOLD or _l16_arith() is None or len(t_shape) != 4:
        return False
    c_out, c_in, kh, kw = weight.shape
    n, _, h, w = t_shape
    key = (n, c_in, c_out, h, w, kh, kw, "act", _l16_arith())
    if key not in _L16_OK:
        _L16_OK[key] = bool(_lib.load().fsc_conv_l16_fwd_act_supported(C.byref(_desc(n, c_in, c_out, h, w, kh, kw, _l16_arith()))))
    return _L16_OK[key]


def conv_l16_act(t, weight, bias, scale, shift, alpha, decl_amax=None, seen=None, prepacked=None):
    """L16 operand of the next convolution = prelu(conv(t) * scale + shift) (fsc_conv_l16_fwd_act).  decl_amax: the declared
    maximum of the result (scaled fp16 limbs only); seen: 1-element tensor that receives max |result| (atomic max; zero it first)."""
    c_out, c_in, kh, kw = weight.shape
    n, _, h, w = t.shape
    d, packed = prepacked if prepacked is not None else conv_l16_pack(weight, n, h, w, False)
    limbs = _l16_limbs()
    out = L16(l16_empty((n, c_out, h, w), weight, limbs), decl_amax if limbs != 3 else None, (n, c_out, h, w), limbs)
    if TIMER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("fsc_conv_l16_fwd_act", C.byref(d), ptr(t.data), ptr(t.amax), ptr(packed), ptr(bias), ptr(scale), ptr(shift), ptr(alpha),
         ptr(out.data), ptr(out.amax), ptr(seen), stream_ptr())
    if TIMER is not None:
        e1.record()
        TIMER.records.append((l16_plan_name(d, 0), 2.0 * n * h * w * c_in * c_out * kh * kw, e0, e1))
        TIMER.note(l16_plan_name(d, 0), 2.0 * n * h * w * c_in * c_out * kh * kw, (n, c_in, c_out, h, w, kh, kw), "fwd")
    return out


def conv_l16_pool(t, weight, bias, prepacked=None, stats_bn=None):
    """3x3 convolution on an L16 operand fused with MaxPool2d(2) (fsc_conv_l16_pool_fwd): (pooled, window index, conv output
    shape), or None when the library has no fused tiling for the shape."""
    c_out, c_in, kh, kw = weight.shape
    n, _, h, w = t.shape
    d = _desc(n, c_in, c_out, h, w, kh, kw, _l16_arith())
    if (kh, kw) != (3, 3) or not _lib.load().fsc_conv_l16_pool_supported(C.byref(d)):
        return None
    _, packed = prepacked if prepacked is not None else conv_l16_pack(weight, n, h, w, False)
    y = torch.empty((n, c_out, h // 2, w // 2), device=weight.device, dtype=torch.float32)
    idx = torch.empty((n, c_out, h // 2, w // 2), device=weight.device, dtype=torch.uint8)
    if TIMER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lay, rec, pivot = _stats_begin(d, True, stats_bn, y)
    if lay is not None:
        call("fsc_conv_l16_pool_fwd_stats", C.byref(d), ptr(t.data), ptr(t.amax), ptr(packed), ptr(bias), ptr(y), ptr(idx),
             ptr(pivot), ptr(rec), stream_ptr())
    else:
        call("fsc_conv_l16_pool_fwd", C.byref(d), ptr(t.data), ptr(t.amax), ptr(packed), ptr(bias), ptr(y), ptr(idx), stream_ptr())
    if TIMER is not None:
        e1.record()
        TIMER.records.append((l16_plan_name(d, 0).replace(">", ",pool>"), 2.0 * n * h * w * c_in * c_out * kh * kw, e0, e1))
        TIMER.note(l16_plan_name(d, 0).replace(">", ",pool>"), 2.0 * n * h * w * c_in * c_out * kh * kw, (n, c_in, c_out, h, w, kh, kw), "fwd")
    if lay is not None:
        _stats_end(lay, rec, y, c_out)
    return y, idx, (n, c_out, h, w)


# conv -> max-pool -> BatchNorm -> PReLU in one launch (inference, fsc_conv_l16_pool_fwd_act): built and bit-identical to the three-pass
# route (tests/test_l3_gpu.py), but OFF by default -- measured on cfg 5 (five fold models on three streams, same box, alternating runs):
# 1151.7 / 1150.8 clips/s without, 1144.5 / 1144.0 with.  The pooled pass and the BatchNorm pass it removes are HBM-bound kernels that
# already hide behind the other streams' matrix-bound convolutions, while the longer epilogue sits on the critical path of this one.
EVAL_POOL_FOLD = os.environ.get("FSC_EVAL_POOL_FOLD", "0") == "1"


def conv_l16_pool_act_supported(t_shape, weight):
    if not EVAL_ACT_FOLD or not EVAL_POOL_FOLD or _l16_arith() not in (9, 10) or len(t_shape) != 4 or tuple(weight.shape[2:]) != (3, 3):
        return False
    c_out, c_in, kh, kw = weight.shape
    n, _, h, w = t_shape
    key = (n, c_in, c_out, h, w, kh, kw, "pool_act", _l16_arith())
    if key not in _L16_OK:
        _L16_OK[key] = bool(_lib.load().fsc_conv_l16_pool_fwd_act_supported(C.byref(_desc(n, c_in, c_out, h, w, kh, kw, _l16_arith()))))
    return _L16_OK[key]


def conv_l16_pool_act(t, weight, bias, scale, shift, alpha, decl_amax=None, seen=None, want_f32=True):
    """Inference: 3x3 convolution -> MaxPool2d(2) -> eval-mode BatchNorm -> PReLU in one launch (fsc_conv_l16_pool_fwd_act,
    reference classifiers.py:526-534): (result as fp32 (N, c_out, H/2, W/2) or None, result as the L16 operand of the next
    convolution).  Neither the convolution output nor the pooled pre-activation is written."""
    c_out, c_in, kh, kw = weight.shape
    n, _, h, w = t.shape
    d, packed = conv_l16_pack(weight, n, h, w, False)
    limbs = _l16_limbs()
    shape = (n, c_out, h // 2, w // 2)
    out = torch.empty(shape, device=weight.device, dtype=torch.float32) if want_f32 else None
    out16 = L16(l16_empty(shape, weight, limbs), decl_amax if limbs != 3 else None, shape, limbs)
    if TIMER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("fsc_conv_l16_pool_fwd_act", C.byref(d), ptr(t.data), ptr(t.amax), ptr(packed), ptr(bias), ptr(scale), ptr(shift), ptr(alpha),
         ptr(out), ptr(out16.data), ptr(out16.amax), ptr(seen), stream_ptr())
    if TIMER is not None:
        e1.record()
        name = l16_plan_name(d, 0).replace(">", ",pool>")
        TIMER.records.append((name, 2.0 * n * h * w * c_in * c_out * kh * kw, e0, e1))
        TIMER.note(name, 2.0 * n * h * w * c_in * c_out * kh * kw, (n, c_in, c_out, h, w, kh, kw), "fwd")
    return out, out16


POOL_FUSION = True


def conv_l16_wgrad_supported(desc):
    return bool(_lib.load().fsc_conv_l16_wgrad_supported(C.byref(desc)))


def l16_wgrad_plan_name(desc):
    buf = C.create_string_buffer(256)
    call("fsc_conv_l16_wgrad_plan_describe", C.byref(desc), buf, 256)
    return buf.value.decode().split(" ")[0]


def conv_l16_wgrad(x16, dout16, weight_shape, out=None):
    """Weight gradient from the L16 input and L16 output gradient of a stride-1 same-pad convolution (into `out` if given).
    With L16_WGRAD_SIDE the kernel runs on the side stream (the caller joins it: ConvBlockFn.backward does)."""
    c_out, c_in, kh, kw = weight_shape
    n, _, h, w = x16.shape
    d = _desc(n, c_in, c_out, h, w, kh, kw, _l16_arith())
    nbytes = _lib.load().fsc_conv_l16_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise _lib.FscError("conv_l16_wgrad: unsupported shape %s" % [getattr(d, f) for f, _ in d._fields_])
    dev = x16.data.device

    def run():
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        dw = out if out is not None else torch.empty(tuple(weight_shape), device=dev, dtype=torch.float32)
        if TIMER is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        call("fsc_conv_l16_wgrad", C.byref(d), ptr(x16.data), ptr(x16.amax), ptr(dout16.data), ptr(dout16.amax), ptr(dw), ptr(ws),
             stream_ptr())
        if TIMER is not None:
            e1.record()
            TIMER.records.append((l16_wgrad_plan_name(d), 2.0 * n * h * w * c_in * c_out * kh * kw, e0, e1))
            TIMER.note(l16_wgrad_plan_name(d), 2.0 * n * h * w * c_in * c_out * kh * kw, (n, c_in, c_out, h, w, kh, kw), "wgrad")
        return dw

    if not L16_WGRAD_SIDE or torch.cuda.is_current_stream_capturing():
        return run()
    if L16_WGRAD_SIDE == 2:
        # beside the BatchNorm passes only: launched on the side stream by the NEXT input-gradient convolution of the backward pass,
        # behind it (l16_wgrad_fork), and joined before the one after that (l16_wgrad_join)
        dw = out if out is not None else torch.empty(tuple(weight_shape), device=dev, dtype=torch.float32)
        pend = _WGRAD.l16_pending

        def deferred(dw=dw):
            side = torch.cuda.current_stream(dev)
            nonlocal out
            out = dw
            run()
            for t in (x16.data, x16.amax, dout16.data, dout16.amax, dw):
                if t is not None:
                    t.record_stream(side)             # keep the allocator from recycling them under the kernel
        pend.append(deferred)
        return dw
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)                    # operands were produced on the main stream
    with torch.cuda.stream(side):
        dw = run()
    for t in (x16.data, x16.amax, dout16.data, dout16.amax):
        if t is not None:
            t.record_stream(side)             # keep the allocator from recycling them under the kernel
    dw.record_stream(main)
    return dw


def l16_wgrad_join(dev):
    """FSC_L16_WGRAD_SIDE=2: the main stream waits for the weight gradients in flight on the side stream (before a convolution is
    launched on the main stream: two matrix-bound kernels never share the chip)."""
    if _WGRAD.l16_inflight:
        torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
        _WGRAD.l16_inflight = False


def l16_wgrad_fork(dev):
    """FSC_L16_WGRAD_SIDE=2: launch the deferred weight gradients on the side stream, behind everything the main stream holds."""
    pend = _WGRAD.l16_pending
    if not pend:
        return
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for fn in pend:
            fn()
    pend.clear()
    _WGRAD.l16_inflight = True


def measure_l16_clock(shape, kind, iters=40):
    """Shader clock (MHz) the chip sustains while it runs the L16 convolution `kind` ("fwd" | "dgrad" | "wgrad") of
    shape (n, c_in, c_out, h, w, kh, kw) back to back on random operands (fsc_conv_l16_last_clock: the kernel stamps the
    shader-cycle and the 100 MHz reference counters itself).  A measurement aid for bench.py; synchronises the device."""
    n, c_in, c_out, h, w, kh, kw = shape
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(n, c_in, h, w, device=dev, generator=gen)
    gy = torch.randn(n, c_out, h, w, device=dev, generator=gen)
    wt = torch.randn(c_out, c_in, kh, kw, device=dev, generator=gen) / (c_in * kh * kw) ** 0.5
    timer, globals()["TIMER"] = TIMER, None
    try:
        if kind == "wgrad":
            x16, g16 = l16_pack(x), l16_pack(gy)
            del x, gy
            for _ in range(iters):
                conv_l16_wgrad(x16, g16, wt.shape)
        else:
            t = l16_pack(gy if kind == "dgrad" else x)
            del x, gy
            pp = conv_l16_pack(wt, n, h, w, kind == "dgrad")
            for _ in range(iters):
                conv_l16(t, wt, None, dgrad=kind == "dgrad", prepacked=pp)
        mhz = C.c_double(0.0)
        call("fsc_conv_l16_last_clock", 1 if kind == "wgrad" else (2 if _l16_limbs() in (3, 4) else 0), C.byref(mhz))
    finally:
        globals()["TIMER"] = timer
    return mhz.value


# FSC_L16_WGRAD_SIDE: where the L16 weight-gradient kernels of the backward pass run.  0: in line.  1: on the side stream as soon as
# their operands exist (round 2, cfg 2, same box: 36.92 -> 36.36 ms per step; two matrix-bound kernels then share the CUs and the
# per-kernel durations that bench.py's roofline and the rocprof tables attribute are no longer those of a kernel alone).  2 (round 5):
# on the side stream BESIDE THE BatchNorm PASSES ONLY -- the gradient of layer k is launched behind layer k's input-gradient
# convolution and joined before layer k-1's, so it overlaps the HBM-bound reduce / apply passes between the two and no two
# convolutions ever run together.
L16_WGRAD_SIDE = int(os.environ.get("FSC_L16_WGRAD_SIDE", "0"))
_L16_OK = {}
USE_L16 = True        # route convolutions through the pre-split (L16) kernels where the library has a tiling for them


def l16_ok(n, c_in, c_out, h, w, kh, kw, dgrad):
    """True when the producer of this convolution's operand should write it as an L16 tensor (split-fp16 arithmetic and
    fsc_conv_l16_fwd has a tiling for the shape)."""
    if not USE_L16 or _l16_arith() is None:
        return False
    key = (n, c_in, c_out, h, w, kh, kw, bool(dgrad), _l16_arith())
    if key not in _L16_OK:
        _L16_OK[key] = conv_l16_supported(_desc(n, c_in, c_out, h, w, kh, kw, _l16_arith()), 1 if dgrad else 0)
    return _L16_OK[key]


def _l16_ok_for(x_shape, weight, dgrad):
    if len(x_shape) != 4:
        return False
    c_out, c_in, kh, kw = weight.shape
    n, h, w = x_shape[0], x_shape[2], x_shape[3]
    return l16_ok(n, c_in, c_out, h, w, kh, kw, dgrad)


def _l16_wgrad_ok_for(x_shape, weight):
    """True when the weight gradient of this convolution runs on fsc_conv_l16_wgrad (both operands as L16 tensors)."""
    if len(x_shape) != 4 or not USE_L16 or _l16_arith() is None:
        return False
    c_out, c_in, kh, kw = weight.shape
    key = (x_shape[0], c_in, c_out, x_shape[2], x_shape[3], kh, kw, "wgrad", _l16_arith())
    if key not in _L16_OK:
        _L16_OK[key] = conv_l16_wgrad_supported(_desc(x_shape[0], c_in, c_out, x_shape[2], x_shape[3], kh, kw, _l16_arith()))
    return _L16_OK[key]


# Weight gradients are MFMA-bound and nothing in the backward chain depends on them until the
# optimizer, while the chain itself alternates MFMA-bound dgrads with HBM-bound BN / pooling
# backward kernels.  Launching wgrad on a second HIP stream lets the two kinds of kernels share the
# chip.  `join_side_stream()` must run before the gradients are consumed.
# Measured on MI355X (cfg 2): +0.8 % step throughput only -- both streams are mostly MFMA-bound, so
# co-running mainly slows each kernel -- and it makes per-kernel event timings meaningless
# (roofline.achieved dropped from 105 to 67 TF for the same work).  Kept as an option, off by default.
ASYNC_WGRAD = False
_SIDE = {}


def _side_stream(device):
    key = str(device)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def join_side_stream(device):
    if str(device) in _SIDE:
        torch.cuda.current_stream(device).wait_stream(_SIDE[str(device)])


def conv_wgrad(x, dout, weight_shape, on_side_stream=False, x_amax=None, dout_amax=None, out=None):
    """Weight gradient.  With on_side_stream the kernel is launched on the side stream and the caller
    must `join_side_stream` before the result is consumed (ConvBlockFn does)."""
    n, c_in, h, w = x.shape
    c_out, _, kh, kw = weight_shape
    d = _desc(n, c_in, c_out, h, w, kh, kw)
    nbytes = _lib.load().fsc_conv_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise _lib.FscError("conv wgrad: unsupported shape")

    def run():
        ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
        dw = out if out is not None else _empty(tuple(weight_shape), x)
        xa, da = _operand_amax(x, x_amax), _operand_amax(dout, dout_amax)
        with _timed(d, 2):
            if _WGRAD.pending is not None and not (ASYNC_WGRAD and on_side_stream):
                # the split-K slices now, their reduce with the block's other weight gradients (wgrad_flush)
                call("fsc_conv_wgrad_partial", C.byref(d), ptr(x), ptr(dout), ptr(ws), ptr(xa), ptr(da), stream_ptr())
                _WGRAD.pending.append((d, ws, dw))
            else:
                call("fsc_conv_wgrad", C.byref(d), ptr(x), ptr(dout), ptr(dw), ptr(ws), ptr(xa), ptr(da), stream_ptr())
        return dw

    if not (ASYNC_WGRAD and on_side_stream):
        return run()
    main = torch.cuda.current_stream(x.device)
    side = _side_stream(x.device)
    side.wait_stream(main)                    # operands were produced on the main stream
    with torch.cuda.stream(side):
        dw = run()
    x.record_stream(side)                     # keep the allocator from recycling them under the kernel
    dout.record_stream(side)
    dw.record_stream(main)
    return dw


# Weight gradients nobody reads before the block's backward returns: between wgrad_begin() and wgrad_flush() conv_wgrad leaves
# its split-K slices in their workspaces, and the flush reduces all of them in one launch (fsc_conv_wgrad_reduce_multi; the four
# reduces of a block of the 1-d model were four launches of ~8 us).  The tensors conv_wgrad returned are complete after the flush.
# The pending list is per THREAD (autograd runs a backward on the thread that called it; two models training on two threads must
# not mix their lists).
class _WgradTLS(threading.local):
    def __init__(self):
        self.pending = None
        self.l16_pending = []          # FSC_L16_WGRAD_SIDE=2: weight gradients waiting for the next input-gradient convolution
        self.l16_inflight = False      # ... and whether the side stream holds some the main stream has not joined


_WGRAD = _WgradTLS()


def wgrad_begin():
    _WGRAD.pending = []


def wgrad_abort():
    _WGRAD.pending = None
    _WGRAD.l16_pending.clear()


def wgrad_flush(end=True):
    """Reduce what is pending (end: and stop deferring)."""
    pending = _WGRAD.pending
    if end:
        _WGRAD.pending = None
    elif pending is not None:
        _WGRAD.pending = []
    if not pending:
        return
    count = len(pending)
    descs = (ConvDesc * count)()
    wsp, dwp = (C.c_void_p * count)(), (C.c_void_p * count)()
    for i, (d, ws, dw) in enumerate(pending):
        descs[i] = d
        wsp[i], dwp[i] = ptr(ws), ptr(dw)
    call("fsc_conv_wgrad_reduce_multi", count, descs, wsp, dwp, stream_ptr())


# ------------------------------------------------------------------------------ BN + PReLU
# BatchNorm workspaces.  With FSC_BN_TICKETS (include/fsc_hip.h) the reduce pass of a call also finalises and counts arrivals in
# the workspace's ticket words, which must be zero when the call is enqueued and are zero again when it has run.  So workspaces
# come from a pool of buffers that were zeroed ONCE, keyed by (device, stream, channels): calls on one stream are ordered, and a
# buffer is handed out again only when its previous holder has let go of it -- two calls that could overlap (different streams: the
# side stream of the weight gradients, SyncBN's communication stream, a second model on its own stream) never share one.  A caller
# gets a LEASE (a view of the pooled buffer); a finaliser on the lease puts the buffer back on the free list when the last reference
# to the lease is gone (explicit check-out / return: no reliance on CPython's reference counts being readable).  The free lists are
# capped; after a failed entry point (_lib.ON_ERROR) the pool is dropped, because an aborted call may have left ticket words non-zero.
# FSC_BN_NO_TICKETS=1 (read once, at import): plain torch.empty workspaces and the separate finalisation launches.
BN_TICKETS = 0 if os.environ.get("FSC_BN_NO_TICKETS") else 32          # FSC_BN_TICKETS
_BN_WS_POOL = {}           # (device, stream, channels) -> free list of zeroed buffers
_BN_WS_GEN = [0]           # bumped by drop_bn_workspaces(): leases of an earlier generation do not come back
_BN_WS_CAP = 16


def _bn_ws_release(key, gen, base):
    if gen == _BN_WS_GEN[0]:
        free = _BN_WS_POOL.setdefault(key, [])
        if len(free) < _BN_WS_CAP:
            free.append(base)


def _bn_ws(c, like):
    nbytes = _lib.load().fsc_bn_workspace_bytes(c)
    if not BN_TICKETS:
        return torch.empty((nbytes + 7) // 8, device=like.device, dtype=torch.float64)
    key = (like.device.index, stream_ptr(), c)
    free = _BN_WS_POOL.get(key)
    base = free.pop() if free else torch.zeros((nbytes + 7) // 8, device=like.device, dtype=torch.float64)
    lease = base.view(-1)                                    # what the caller holds (and may keep across both phases of a call)
    weakref.finalize(lease, _bn_ws_release, key, _BN_WS_GEN[0], base)
    return lease


def drop_bn_workspaces():
    """Forget the pooled BatchNorm workspaces (after a device error: their ticket words may be left non-zero).  Buffers that are
    leased out at this moment are not taken back either."""
    _BN_WS_GEN[0] += 1
    _BN_WS_POOL.clear()


_lib.ON_ERROR.append(drop_bn_workspaces)


class BNState:
    """What one fused BN(+residual)(+PReLU) unit keeps between forward and backward."""
    __slots__ = ("mean", "invstd", "scale", "shift", "minmax", "pending")

    def __init__(self):
        self.pending = None         # bn_prepare(lazy=True): the statistics call, not issued yet (see _bn_realize)


def _sync_buffer(c, like):
    return torch.empty(4 * c, device=like.device, dtype=torch.float64)      # FSC_BN_SYNC_DOUBLES(c)


FUSE_OUT_STATS = True          # the block's last unit reduces the next BatchNorm's statistics and the global max itself
_STATS_FOLDED = 4              # FSC_BN_STATS_FOLDED
_PRESTATS = {}                 # at most one entry: (data_ptr, shape, version) of a block output -> its folded BN workspace


def _take_prestats(x):
    """The BatchNorm workspace the producer of `x` filled (bn_act_forward_rec), if x is that very tensor, unmodified."""
    if not _PRESTATS:
        return None
    key, (ws, _alive, flags) = _PRESTATS.popitem()   # (the entry held the tensor, so its address was not reused meanwhile)
    if key == (x.data_ptr(), tuple(x.shape), x._version, x.device):
        return ws, flags
    return None


def bn_act_forward_unit(x, st, alpha, residual, want_gmax):
    """A lazily prepared unit whose channel is one workgroup (bn_prepare(lazy=True) left st.pending): statistics, apply pass and --
    where the planes allow -- the global max-pool of the output in ONE launch.  Returns (y, feat, fidx), or None when this
    shape cannot serve the global max that way (the caller then takes bn_act_forward_rec)."""
    pend = st.pending
    if pend is None or pend[0] is not x:
        return None
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    if want_gmax and not (_lib.load().fsc_bn_train_act_fwd_supported(n, c, hw) & 2):
        return None
    (_x, a), st.pending = pend, None
    y = torch.empty_like(x)
    feat = _empty((n, c), x) if want_gmax else None
    fidx = _empty((n, c), x, torch.int32) if want_gmax else None
    with _stage("bn_act_fwd", 2 * _nb(x) + _nb(residual, y)):
        call("fsc_bn_train_act_fwd", a[0], ptr(residual), *a[1:], ptr(st.minmax), ptr(alpha), ptr(y), ptr(feat), ptr(fidx),
             stream_ptr())
    return y, feat, fidx


def bn_act_forward_rec(x, st, alpha, residual, want_stats, want_gmax):
    """bn_act_forward (fp32 output) that also reduces, while it writes y, what the next readers of y need: the statistics
    partials of the BatchNorm that follows (kept for the bn_prepare call on this very tensor) and the global max-pool.
    Returns (y, feat or None, feat_idx or None)."""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    _bn_realize(st)
    y = torch.empty_like(x)
    rec = torch.empty((_lib.load().fsc_bn_records_bytes(n, c, hw) + 7) // 8, device=x.device, dtype=torch.float64)
    with _stage("bn_act_fwd", _nb(x, residual, y)):
        call("fsc_bn_act_fwd_rec", ptr(x), ptr(residual), ptr(st.scale), ptr(st.shift), ptr(alpha), ptr(y), n, c, hw, ptr(rec),
             stream_ptr())
    ws = _bn_ws(c, x) if want_stats else None
    feat = _empty((n, c), x) if want_gmax else None
    fidx = _empty((n, c), x, torch.int32) if want_gmax else None
    call("fsc_bn_records_fold", ptr(rec), ptr(y), n, c, hw, ptr(ws), ptr(feat), ptr(fidx), stream_ptr())
    _PRESTATS.clear()
    if want_stats:
        _PRESTATS[(y.data_ptr(), tuple(y.shape), y._version, y.device)] = (ws, y, _STATS_FOLDED)
    return y, feat, fidx


def _bn_eval_key(bn, gamma, beta):
    return tuple((t.data_ptr(), t._version) if t is not None else None
                 for t in (gamma, beta, bn.running_mean, bn.running_var)) + (bn.eps,)


def _bn_eval_scale_shift(bn, gamma, beta, scale=None, shift=None):
    """Folded scale / shift of an eval-mode BatchNorm into (or instead of) the given buffers; cached across batches under no_grad."""
    key = _bn_eval_key(bn, gamma, beta) if not torch.is_grad_enabled() else None
    hit = _EVAL_BN.get(key) if key is not None else None
    if hit is not None:
        return hit[0], hit[1]
    if scale is None:
        scale, shift = _empty((bn.running_mean.numel(),), bn.running_mean), _empty((bn.running_mean.numel(),), bn.running_mean)
    call("fsc_bn_eval_prepare", scale.numel(), ptr(gamma), ptr(beta), ptr(bn.running_mean), ptr(bn.running_var),
         bn.eps, ptr(scale), ptr(shift), stream_ptr())
    if key is not None:
        if len(_EVAL_BN) > 4096:
            _EVAL_BN.clear()
        _EVAL_BN[key] = (scale, shift, bn)          # (the module is kept alive: its addresses cannot be reused)
    return scale, shift


BN_FUSED = os.environ.get("FSC_BN_FUSED", "1") != "0"


def _bn_realize(st):
    """Issue the statistics call a lazy bn_prepare held back (a consumer other than the plain apply pass needs them)."""
    pend, st.pending = st.pending, None
    if pend is not None:
        x, args = pend
        with _stage("bn_stats", _nb(x)):
            call("fsc_bn_train_stats", *args, ptr(_bn_ws(x.shape[1], x)), None, BN_TICKETS, ptr(st.minmax), stream_ptr())


def bn_prepare(x, bn, training, sync=None, defer=None, want_minmax=False, lazy=False):
    """Batch statistics (training; also updates the running stats, once) or running statistics
    (eval) -> per-channel scale/shift.  `sync` (a callable that sum-all-reduces a device tensor in place over the
    data-parallel replicas, see parallel.SyncBN) turns the batch statistics into cross-replica statistics.
    lazy: the caller hands `st` to bn_act_forward / bn_act_forward_rec next -- where the library runs statistics and apply pass
    of such a shape in one launch (fsc_bn_train_act_fwd: a channel is one workgroup) the statistics are left to that call."""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    st = BNState()
    st.scale = _empty((c,), x)
    st.shift = _empty((c,), x)
    st.minmax = None
    gamma, beta = bn.weight, bn.bias
    if training or bn.running_mean is None:
        if _EVAL_BN:
            _EVAL_BN.clear()                                # (running statistics are about to move under raw pointers)
        st.mean = _empty((c,), x)
        st.invstd = _empty((c,), x)
        track = training and bn.track_running_stats and bn.running_mean is not None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        if track:
            if bn.momentum is not None and (defer is not None or _COUNTER_SINK is not None):
                # (bumped later: by the caller for its block, or by counters_flush for the whole forward, in one launch)
                (defer if defer is not None else _COUNTER_SINK).append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
            if bn.momentum is None:
                momentum = 1.0 / float(bn.num_batches_tracked)
        pre = _take_prestats(x)                            # reduced by the kernel that wrote x?
        if pre is not None and (pre[1] & _STATS_PIVOT_RM) and not training:
            pre = None                                     # (sums about a running mean that this call would not pass: never in practice)
        st.minmax = _empty((2 * c,), x)                    # per-channel [min, max] of x: the L16 producers' bound
        if pre is not None and (pre[1] & _STATS_CONV_REC):
            if sync is None and hw > 1:                    # fold + finalise in one launch
                rec, lay, _c = pre[0]
                with _stage("bn_stats", 0):
                    call("fsc_bn_train_stats_conv", ptr(rec), lay[0], lay[1], lay[2], lay[3], ptr(x), n, c, hw, ptr(gamma), ptr(beta),
                         bn.eps, momentum, ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                         ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), ptr(st.minmax), stream_ptr())
                return st
            pre = (_fold_conv_records(pre[0], x), pre[1] & ~_STATS_CONV_REC)
        if (lazy and BN_FUSED and pre is None and sync is None and hw > 1
                and _lib.load().fsc_bn_train_act_fwd_supported(n, c, hw)):
            st.pending = (x, (ptr(x), n, c, hw, ptr(gamma), ptr(beta), bn.eps, momentum,
                              ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                              ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift)))
            return st
        ws, folded = pre if pre is not None else (_bn_ws(c, x), 0)      # (kept alive across both phases)
        args = (ptr(x), n, c, hw, ptr(gamma), ptr(beta), bn.eps, momentum,
                ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), ptr(ws))
        if sync is None:
            with _stage("bn_stats", 0 if folded else _nb(x)):      # (folded: the producer of x reduced them; only the finalisation runs)
                call("fsc_bn_train_stats", *args, None, folded if pre is not None else BN_TICKETS, ptr(st.minmax), stream_ptr())
        else:
            moments = _sync_buffer(c, x)                   # [sum x, sum x^2, count, 0] per channel, fp64
            call("fsc_bn_train_stats", *args, ptr(moments), 1 | folded, ptr(st.minmax), stream_ptr())
            sync(moments)
            call("fsc_bn_train_stats", *args, ptr(moments), 2, ptr(st.minmax), stream_ptr())
    else:
        st.mean = None
        st.invstd = None
        # the folded scale / shift of an eval-mode BatchNorm are reused across batches (keyed like the packed weights: addresses
        # and versions; a training forward or an optimizer step of this library drops the cache)
        st.scale, st.shift = _bn_eval_scale_shift(bn, gamma, beta, st.scale, st.shift)
        pre = _take_prestats(x)
        if pre is not None and (pre[1] & _STATS_CONV_REC):
            if want_minmax and EVAL_L16 and hw > 1:          # min / max straight from the conv epilogue's records, one launch
                rec, lay, _c = pre[0]
                st.minmax = _empty((2 * c,), x)
                call("fsc_bn_train_stats_conv", ptr(rec), lay[0], lay[1], lay[2], lay[3], ptr(x), n, c, hw, None, None, bn.eps, 0.0,
                     None, None, None, None, None, None, ptr(st.minmax), stream_ptr())
                return st
            pre = None
        if want_minmax and EVAL_L16 and hw > 1:
            # inference on the L16 kernels: the producer that writes the conv operand needs the range of x up front -- from the
            # records of the kernel that wrote x, else from one reduction pass
            ws, folded = pre if pre is not None else (_bn_ws(c, x), 0)
            st.minmax = _empty((2 * c,), x)
            call("fsc_bn_train_stats", ptr(x), n, c, hw, None, None, bn.eps, 0.0, None, None, None, None, None, None, ptr(ws),
                 None, ((folded & _STATS_FOLDED) if pre is not None else BN_TICKETS) | _STATS_MINMAX_ONLY, ptr(st.minmax), stream_ptr())
    return st


def _want_amax():
    return get_conv_arith() == 3


def bn_act_forward(x, st, alpha=None, residual=None, with_amax=False, l16=False, want_f32=True):
    """y = act(bn(x) [+ residual]).  with_amax: returns (y, max |y| as a device scalar or None) -- the scale
    of the split-fp16 conv kernels, reported by the kernel that writes y.
    l16: returns (y or None, L16 or None) instead -- y also (want_f32) or only as a pre-split L16 tensor for
    conv_l16, when the unit has batch statistics (their min / max bound the output up front) and no residual."""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    limbs = _l16_limbs()
    to_l16 = l16 and (st.minmax is not None or limbs == 3) and residual is None and hw > 1
    if st.pending is not None:
        if not to_l16 and not ((with_amax or l16) and _want_amax()) and st.pending[0] is x:
            # statistics + apply in one launch (a channel is one workgroup)
            (_x, a), st.pending = st.pending, None
            y = torch.empty_like(x)
            with _stage("bn_act_fwd", 2 * _nb(x) + _nb(residual, y)):
                call("fsc_bn_train_act_fwd", a[0], ptr(residual), *a[1:], ptr(st.minmax), ptr(alpha), ptr(y), None, None, stream_ptr())
            if l16:
                return y, None
            return (y, None) if with_amax else y
        _bn_realize(st)
    if to_l16:
        y = torch.empty_like(x) if want_f32 else None
        t = L16(l16_empty(x.shape, x, limbs), _empty((AMAX_FLOATS,), x) if limbs != 3 else None, x.shape, limbs)
        with _stage("bn_act_fwd", _nb(x, y, t)):
            call("fsc_bn_act_fwd_limbs", ptr(x), None, ptr(st.scale), ptr(st.shift), ptr(alpha), ptr(y),
                 n, c, hw, ptr(t.amax), ptr(st.minmax) if limbs != 3 else None, ptr(t.data), limbs, stream_ptr())
        return y, t
    y = torch.empty_like(x)
    y_amax = _empty((AMAX_FLOATS,), x) if (with_amax or l16) and _want_amax() else None
    with _stage("bn_act_fwd", _nb(x, residual, y)):
        call("fsc_bn_act_fwd", ptr(x), ptr(residual), ptr(st.scale), ptr(st.shift), ptr(alpha), ptr(y),
             n, c, hw, ptr(y_amax), None, None, stream_ptr())
    if l16:
        return y, None
    return (y, y_amax) if with_amax else y


def bn_act_backward(dy, x, st, bn, alpha=None, residual=None, gmax=None, want_dx=True,
                    want_dres=False, want_chan_sum=False, with_amax=False, sync=None, l16=False, want_f32=True):
    """Returns (dx, dresidual, dgamma, dbeta, dalpha, dx_chan_sum) [+ (max |dx|,) with with_amax].
    want_dx=False: only the parameter gradients (dx, dresidual, dx_chan_sum and the amax come back as None).
    l16 (with with_amax): dx is also (want_f32) or only written as an L16 tensor; the last element of the result is then
    that L16 (its .amax is the declared bound) instead of the amax buffer."""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    l16 = l16 and with_amax and hw > 1 and want_dx
    dx = torch.empty_like(x) if (want_f32 or not l16) and want_dx else None
    dres = torch.empty_like(x) if want_dres and want_dx else None
    dgamma = _empty((c,), x)
    dbeta = _empty((c,), x)
    dalpha = _empty((c,), x) if alpha is not None else None
    csum = _empty((c,), x) if want_chan_sum and want_dx else None
    gdy, gidx = gmax if gmax is not None else (None, None)
    limbs = _l16_limbs()
    lflag = _BN_L16_FLAG.get(limbs, 0) if l16 else 0
    dx_amax = _empty((AMAX_FLOATS,), x) if with_amax and want_dx and (_want_amax() or (l16 and limbs != 3)) else None
    t = L16(l16_empty(x.shape, x, limbs), dx_amax, x.shape, limbs) if l16 else None
    ws = _bn_ws(c, x)                                      # (kept alive across both phases)
    args = (ptr(dy), ptr(gdy), ptr(gidx), ptr(x), ptr(residual), ptr(st.mean),
            ptr(st.invstd), ptr(bn.weight), ptr(bn.bias), ptr(alpha), ptr(dx), ptr(dres), ptr(dgamma),
            ptr(dbeta), ptr(dalpha), ptr(csum), n, c, hw, ptr(ws), ptr(dx_amax))
    t_ptr = ptr(t.data) if l16 else None
    if not want_dx:
        # parameter gradients only (an input that needs no gradient): the reduce pass and its finalisation -- phase 1 of the
        # replicated form, whose sums (this replica's; the gradient all-reduce adds the replicas) nobody reads here
        with _stage("bn_act_bwd", _nb(dy, x, residual)):
            call("fsc_bn_act_bwd", *args, ptr(_sync_buffer(c, x)), 1, None, stream_ptr())
    elif sync is None:
        # reduce pass reads dy, x, residual; apply pass reads them again and writes dx (fp32 and / or L16) and dresidual
        with _stage("bn_act_bwd", 2 * _nb(dy, x, residual) + _nb(dx, dres, t)):
            call("fsc_bn_act_bwd", *args, None, BN_TICKETS | lflag, t_ptr, stream_ptr())
    else:
        sums = _sync_buffer(c, x)                          # [sum dz, sum dz * xhat, count, 0] per channel
        call("fsc_bn_act_bwd", *args, ptr(sums), 1 | lflag, t_ptr, stream_ptr())
        sync(sums)
        call("fsc_bn_act_bwd", *args, ptr(sums), 2 | lflag, t_ptr, stream_ptr())
    if with_amax:
        return dx, dres, dgamma, dbeta, dalpha, csum, (t if l16 else dx_amax)
    return dx, dres, dgamma, dbeta, dalpha, csum


def bn_act_backward_unpool(dy, x, st, bn, alpha, pool_idx, c_shape, ph, sync=None, l16=False, want_f32=True):
    """Backward of BN+PReLU on a pooled tensor fused with the max-pool backward.
    Returns (dc at the un-pooled shape, dgamma, dbeta, dalpha, per-channel sum of the gradient, max |dc| or None); with
    l16 the last element is dc as an L16 tensor (and dc itself None unless want_f32)."""
    n, c, h, w = c_shape
    dc = _empty(tuple(c_shape), x) if (want_f32 or not l16) else None
    dgamma = _empty((c,), x)
    dbeta = _empty((c,), x)
    dalpha = _empty((c,), x) if alpha is not None else None
    csum = _empty((c,), x)
    limbs = _l16_limbs()
    lflag = _BN_L16_FLAG.get(limbs, 0) if l16 else 0
    dc_amax = _empty((AMAX_FLOATS,), x) if (_want_amax() or (l16 and limbs != 3)) else None
    t = L16(l16_empty(c_shape, x, limbs), dc_amax, c_shape, limbs) if l16 else None
    ws = _bn_ws(c, x)                                      # (kept alive across both phases)
    args = (ptr(dy), ptr(x), ptr(st.mean), ptr(st.invstd), ptr(bn.weight), ptr(bn.bias),
            ptr(alpha), ptr(pool_idx), ptr(dc), ptr(dgamma), ptr(dbeta), ptr(dalpha), ptr(csum), n, c, h, w, ph,
            ptr(ws), ptr(dc_amax))
    t_ptr = ptr(t.data) if l16 else None
    if sync is None:
        with _stage("bn_act_bwd", 2 * _nb(dy, x) + _nb(pool_idx, dc, t)):
            call("fsc_bn_act_bwd_unpool", *args, None, BN_TICKETS | lflag, t_ptr, stream_ptr())
    else:
        sums = _sync_buffer(c, x)
        call("fsc_bn_act_bwd_unpool", *args, ptr(sums), 1 | lflag, t_ptr, stream_ptr())
        sync(sums)
        call("fsc_bn_act_bwd_unpool", *args, ptr(sums), 2 | lflag, t_ptr, stream_ptr())
    return dc, dgamma, dbeta, dalpha, csum, (t if l16 else dc_amax)


# ------------------------------------------------------------------------------ pooling
def maxpool_forward(x, ph):
    n, c, h, w = x.shape
    y = _empty((n, c, h // ph, w // 2), x)
    idx = _empty((n, c, h // ph, w // 2), x, torch.uint8)
    with _stage("pool", _nb(x, y, idx)):
        call("fsc_maxpool_fwd", ptr(x), ptr(y), ptr(idx), n * c, h, w, ph, stream_ptr())
    return y, idx


def maxpool_backward(dy, idx, x_shape, ph):
    n, c, h, w = x_shape
    dx = _empty(tuple(x_shape), dy)
    call("fsc_maxpool_bwd", ptr(dy), ptr(idx), ptr(dx), n * c, h, w, ph, stream_ptr())
    return dx


def global_maxpool_forward(x):
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    y = _empty((n, c), x)
    idx = _empty((n, c), x, torch.int32)
    with _stage("pool", _nb(x, y, idx)):
        call("fsc_global_maxpool_fwd", ptr(x), ptr(y), ptr(idx), n * c, hw, stream_ptr())
    return y, idx


# ------------------------------------------------------------------------------ RNN aggregation head
def _raw_call_ptr(t, offset_floats):
    """Device address `offset_floats` into a contiguous tensor (a time slice of a (N, T, G) buffer)."""
    return t.data_ptr() + 4 * offset_floats


class RnnHeadFn(torch.autograd.Function):
    """aggregation_type == "rnn" head of one block (reference networks/classifiers.py:514-522, 592-597; 1-d :137-145,
    202-207): mean over frequency, LayerNorm((C,)), bidirectional GRU(C, hidden), features = the two final states.
    Input projections and weight gradients are GEMMs (fsc_linear_*), the recurrence is one fsc_gru_step_* per step."""

    @staticmethod
    def forward(ctx, out, eps, ln_w, ln_b, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        out = out.contiguous()
        n, c, h, w = out.shape
        hid = w_hh.shape[1]
        keep = any(ctx.needs_input_grad)
        xm = _empty((n, w, c), out)
        call("fsc_freq_mean_fwd", ptr(out), ptr(xm), n, c, h, w, stream_ptr())
        rows = n * w
        xl = torch.empty_like(xm)
        mean, rstd = _empty((rows,), out), _empty((rows,), out)
        call("fsc_layernorm_fwd", ptr(xm), ptr(ln_w), ptr(ln_b), float(eps), ptr(xl), ptr(mean), ptr(rstd), rows, c, stream_ptr())
        feats, saved = [], []
        for rev, (wi, wh, bi, bh) in enumerate(((w_ih, w_hh, b_ih, b_hh), (w_ih_r, w_hh_r, b_ih_r, b_hh_r))):
            gx = _empty((n, w, 3 * hid), out)
            call("fsc_linear_fwd", ptr(xl), ptr(wi), ptr(bi), ptr(gx), rows, c, 3 * hid, stream_ptr())
            hs = _empty((w + 1, n, hid), out)
            call("fsc_fill", ptr(hs[0]), 0.0, n * hid, stream_ptr())
            gates = _empty((4, w, n, hid), out) if keep else None
            for s in range(w):
                t = w - 1 - s if rev else s
                g = [ptr(gates[q, s]) for q in range(4)] if keep else [None] * 4
                call("fsc_gru_step_fwd", _raw_call_ptr(gx, t * 3 * hid), w * 3 * hid, ptr(hs[s]), ptr(wh), ptr(bh),
                     ptr(hs[s + 1]), g[0], g[1], g[2], g[3], n, hid, stream_ptr())
            feats.append(hs[w])
            saved.append((hs, gates))
        feat = torch.cat(feats, dim=1)
        if keep:
            ctx.save_for_backward(xm, xl, mean, rstd, ln_w, w_ih, w_hh, w_ih_r, w_hh_r)
            ctx.saved_steps = saved
            ctx.shape = (n, c, h, w, hid)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        xm, xl, mean, rstd, ln_w, w_ih, w_hh, w_ih_r, w_hh_r = ctx.saved_tensors
        n, c, h, w, hid = ctx.shape
        rows = n * w
        dfeat = dfeat.contiguous()
        grads = []
        dxl = None
        for rev, (wi, wh) in enumerate(((w_ih, w_hh), (w_ih_r, w_hh_r))):
            hs, gates = ctx.saved_steps[rev]
            dgx = _empty((n, w, 3 * hid), xm)
            dgh = _empty((w, n, 3 * hid), xm)
            dh = dfeat[:, rev * hid:(rev + 1) * hid].contiguous()
            dh_prev = torch.empty_like(dh)
            for s in range(w - 1, -1, -1):
                t = w - 1 - s if rev else s
                call("fsc_gru_step_bwd", ptr(dh), ptr(gates[0, s]), ptr(gates[1, s]), ptr(gates[2, s]), ptr(gates[3, s]),
                     ptr(hs[s]), ptr(wh), _raw_call_ptr(dgx, t * 3 * hid), w * 3 * hid, ptr(dgh[s]), ptr(dh_prev), n, hid,
                     stream_ptr())
                dh, dh_prev = dh_prev, dh
            d_wh, d_bh = torch.empty_like(wh), _empty((3 * hid,), xm)
            call("fsc_linear_bwd", ptr(dgh), ptr(hs[:w]), ptr(wh), None, ptr(d_wh), ptr(d_bh), w * n, hid, 3 * hid, stream_ptr())
            d_wi, d_bi = torch.empty_like(wi), _empty((3 * hid,), xm)
            dx_dir = torch.empty_like(xl)
            call("fsc_linear_bwd", ptr(dgx), ptr(xl), ptr(wi), ptr(dx_dir), ptr(d_wi), ptr(d_bi), rows, c, 3 * hid, stream_ptr())
            if dxl is None:
                dxl = dx_dir
            else:
                call("fsc_axpy", ptr(dx_dir), 1.0, ptr(dxl), dxl.numel(), stream_ptr())
            grads += [d_wi, d_wh, d_bi, d_bh]
        dxm = torch.empty_like(xm)
        d_lnw, d_lnb = _empty((c,), xm), _empty((c,), xm)
        call("fsc_layernorm_bwd", ptr(dxl), ptr(xm), ptr(mean), ptr(rstd), ptr(ln_w), ptr(dxm), ptr(d_lnw), ptr(d_lnb), rows, c,
             stream_ptr())
        dout = None
        if ctx.needs_input_grad[0]:
            dout = _empty((n, c, h, w), xm)
            call("fsc_freq_mean_bwd", ptr(dxm), ptr(dout), n, c, h, w, stream_ptr())
        ctx.saved_steps = None
        return (dout, None, d_lnw, d_lnb) + tuple(grads)


def rnn_head(out, rnn):
    """`rnn` = the reference's nn.Sequential(nn.LayerNorm((C,)), nn.GRU(C, 128, batch_first=True, bidirectional=True))
    parameter holder; `out` (N, C, H, W) block output (H == 1 for the 1-d model).  Returns (N, 2 * hidden)."""
    _need_cuda(out, "rnn_head")
    ln, gru = rnn[0], rnn[1]
    return RnnHeadFn.apply(out, ln.eps, ln.weight, ln.bias, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0,
                           gru.weight_ih_l0_reverse, gru.weight_hh_l0_reverse, gru.bias_ih_l0_reverse, gru.bias_hh_l0_reverse)


# ------------------------------------------------------------------------------ conv block
def _conv_params(conv):
    """(weight as 4-d, bias, kh, kw) for nn.Conv2d / nn.Conv1d parameter holders."""
    w = conv.weight
    if w.dim() == 3:
        w = w.unsqueeze(2)
    return w, conv.bias


class _BlockCtx:
    pass


def _bn_fwd_for_conv(x, st, alpha, weight, keep_f32=False, need_wgrad=True):
    """BN (+ PReLU) output that feeds the convolution `weight`: (y fp32 or None, max |y| buffer or None, y as L16 or None).
    The L16 form is written when the forward convolution or its weight gradient reads it; the fp32 form when one of the two
    does not (or the caller needs it: keep_f32)."""
    fw, wg = _l16_ok_for(x.shape, weight, False), need_wgrad and _l16_wgrad_ok_for(x.shape, weight)
    if (fw or wg) and (st.minmax is not None or _l16_limbs() == 3):
        y, t = bn_act_forward(x, st, alpha, l16=True, want_f32=keep_f32 or not (fw and (wg or not need_wgrad)))
        if t is not None:
            return y, t.amax, t
    y, y_max = bn_act_forward(x, st, alpha, with_amax=True)
    return y, y_max, None


def _conv_fwd_any(x, x_16, weight, bias, x_amax, packs=None, stats_bn=None):
    """`packs` (a list, training only): the input-gradient fragments of this weight are packed in the same call as the
    forward ones and appended for the backward pass (or None when that direction does not run on the L16 kernels)."""
    if x_16 is not None and _l16_ok_for(x_16.shape, weight, False):
        if packs is not None:
            n, _, h, w = x_16.shape
            pf, pd = conv_l16_pack_pair(weight, n, h, w)
            packs.append(pd)
            return conv_l16(x_16, weight, bias, prepacked=pf, stats_bn=stats_bn)
        return conv_l16(x_16, weight, bias, stats_bn=stats_bn)
    if packs is not None:
        packs.append(None)
    return conv_forward(x, weight, bias, x_amax=x_amax, stats_bn=stats_bn)


def _conv_dgrad_any(dout, dout_16, weight, x_shape, dout_amax, accumulate_into=None, prepacked=None):
    side2 = L16_WGRAD_SIDE == 2 and (_WGRAD.l16_inflight or _WGRAD.l16_pending)
    if side2:
        l16_wgrad_join(dout.device if dout is not None else dout_16.data.device)
    if dout_16 is not None and _l16_ok_for(x_shape, weight, True):
        r = conv_l16(dout_16, weight, None, dgrad=True, accumulate_into=accumulate_into, prepacked=prepacked)
    else:
        r = conv_dgrad(dout, weight, x_shape, accumulate_into=accumulate_into, dout_amax=dout_amax)
    if side2:
        l16_wgrad_fork(r.device)
    return r


GRAD_OUT = None       # callable(weight) -> tensor to write that weight's gradient into, or None (parallel.BucketedGradReducer.grad_view)


def _conv_wgrad_any(x, x_16, x_amax, dout, dout_16, dout_amax, weight):
    out = GRAD_OUT(weight) if GRAD_OUT is not None else None        # data-parallel: straight into the all-reduce bucket
    if x_16 is not None and dout_16 is not None and _l16_wgrad_ok_for(x_16.shape, weight):
        return conv_l16_wgrad(x_16, dout_16, weight.shape, out=out)
    return conv_wgrad(x, dout, weight.shape, True, x_amax=x_amax, dout_amax=dout_amax, out=out)


def _grad_formats(x_shape, weight):
    """(write the gradient of this convolution's output as L16, keep its fp32 form): L16 when the input-gradient kernel or
    the weight-gradient kernel reads it, fp32 when one of them does not."""
    dg, wg = _l16_ok_for(x_shape, weight, True), _l16_wgrad_ok_for(x_shape, weight)
    return (dg or wg), not (dg and wg)


def _amax_of(m):
    """The amax buffer inside what the backward BN kernels return as their last element (an L16 or the buffer itself)."""
    return m.amax if isinstance(m, L16) else m


def _l16_of(m):
    return m if isinstance(m, L16) else None


def _act_key(bn, weight, bias=None, alpha=None):
    """Everything the output range of conv -> eval BatchNorm -> PReLU depends on besides the input: (address, version) of the
    BatchNorm's parameters / buffers, the convolution's weight and bias and the PReLU slope."""
    extra = tuple((t.data_ptr(), t._version) if t is not None else (0, 0) for t in (weight, bias, alpha))
    return _bn_eval_key(bn, bn.weight, bn.bias) + extra


def _eval_conv_act(t16, weight, bias, bn, alpha, next_weight):
    """Inference: conv -> eval-mode BatchNorm -> PReLU in ONE launch that writes the next convolution's L16 operand
    (conv_l16_act), or None when the layer takes the two-pass route (no tiling, the next convolution reads fp32, or -- scaled
    fp16 limbs -- no scope / no calibration yet)."""
    if not EVAL_ACT_FOLD or t16 is None or torch.is_grad_enabled() or bn.running_mean is None or len(t16.shape) != 4:
        return None
    n, _, h, w = t16.shape
    if not conv_l16_act_supported(t16.shape, weight) or not _l16_ok_for((n, weight.shape[0], h, w), next_weight, False):
        return None
    decl = seen = None
    if _l16_limbs() != 3:                                    # (scaled fp16 limbs: f16x3, f16x6)
        scope = _ACT_TLS.scope
        key = _act_key(bn, weight, bias, alpha)
        cal = _ACT_CAL.get(key) if scope is not None else None
        if cal is None:
            return None
        decl = cal[0]
        seen = scope.slot(weight, cal[1], key)
    scale, shift = _bn_eval_scale_shift(bn, bn.weight, bn.bias)
    return conv_l16_act(t16, weight, bias, scale, shift, alpha, decl, seen)


def _eval_conv_pool_act(a16, weight, bias, bn, alpha, next_weight):
    """Inference: conv3x3 -> max-pool -> eval-mode BatchNorm -> PReLU as ONE launch writing fp32 (for the residual) and the next
    convolution's limbs (conv_l16_pool_act), or None (two-pass route: no tiling, the unit's first convolution reads fp32, or --
    scaled fp16 limbs -- no scope / no calibration yet)."""
    if (not EVAL_ACT_FOLD or a16 is None or torch.is_grad_enabled() or bn.running_mean is None or len(a16.shape) != 4
            or not POOL_FUSION):
        return None
    n, _, h, w = a16.shape
    if not conv_l16_pool_act_supported(a16.shape, weight) or not _l16_ok_for((n, weight.shape[0], h // 2, w // 2), next_weight, False):
        return None
    decl = seen = None
    if _l16_limbs() != 3:
        scope = _ACT_TLS.scope
        key = _act_key(bn, weight, bias, alpha)
        cal = _ACT_CAL.get(key) if scope is not None else None
        if cal is None:
            return None
        decl = cal[0]
        seen = scope.slot(weight, cal[1], key)
    scale, shift = _bn_eval_scale_shift(bn, bn.weight, bn.bias)
    return conv_l16_pool_act(a16, weight, bias, scale, shift, alpha, decl, seen)


def _act_calibrate(bn, weight, t16, bias=None, alpha=None):
    """Scaled fp16 limbs, inference on the two-pass route: remember TWICE the bound this batch declared for the layer's output as
    the scale of the folded launches to come (the fp16 high limb itself holds another factor 1.99 above the declared maximum:
    `limit`, what ActFoldScope.ok() compares the largest value written with)."""
    if not EVAL_ACT_FOLD or _l16_arith() is None or _l16_limbs() == 3 or t16 is None or t16.amax is None or torch.is_grad_enabled():
        return
    key = _act_key(bn, weight, bias, alpha)
    if key in _ACT_CAL:
        return
    if len(_ACT_CAL) > 4096:
        _ACT_CAL.clear()
    decl = t16.amax * 2.0
    _ACT_CAL[key] = (decl, (decl.max() * 1.99).reshape(1), bn, weight, bias, alpha)   # (kept alive: their addresses cannot be reused)


def _block_forward(x, mods, training, want_head, ph, keep, sync=None, next_bn=False):
    """BN -> conv3 -> maxpool -> BN+PReLU -> residual unit (-> global max).  `mods` is the
    reference's nn.Sequential of parameter holders.  Returns (out, feat, ctx)."""
    bn_a, conv_a, _pool, bn_b, prelu_b, res = mods[0], mods[1], mods[2], mods[3], mods[4], mods[5]
    k = _BlockCtx()
    counters = []                      # num_batches_tracked of the five BatchNorms: one launch at the end
    k.x_shape = tuple(x.shape)
    w_a, b_a = _conv_params(conv_a)

    def mm(t, wt):           # inference: the BatchNorm in front of a convolution with an L16 tiling also reports the range of its input
        return (not training) and EVAL_L16 and _l16_limbs() != 3 and t.dim() == 4 and _l16_ok_for(t.shape, wt, False)   # (bf16 limbs: no scale, no range)

    st_a = bn_prepare(x, bn_a, training, sync, counters, want_minmax=mm(x, w_a), lazy=True)
    # Operands of convolutions that have an L16 tiling are written pre-split by the BN / PReLU kernel that produces them
    # (`*_16`); the fp32 copy stays for the weight gradient (and, for b, the residual).
    a, a_max, a_16 = _bn_fwd_for_conv(x, st_a, None, w_a, need_wgrad=keep)
    packs = [] if keep else None          # input-gradient weight fragments packed along with the forward ones
    w1, b1 = _conv_params(res.conv1)
    # inference: conv_a -> max-pool -> bn_b -> PReLU in one launch (the pooled pre-activation never exists); else the steps below
    pool_act = _eval_conv_pool_act(a_16, w_a, b_a, bn_b, prelu_b.weight, w1) if (ph == 2 and not training and not keep) else None
    if pool_act is not None:
        b, b_16 = pool_act
        b_max = b_16.amax
        st_b = None
        k.c_shape = (a_16.shape[0], w_a.shape[0], a_16.shape[2], a_16.shape[3])
    fused = conv_pool_forward(a, w_a, b_a) if (ph == 2 and a is not None and pool_act is None) else None
    if fused is None and ph == 1 and a is not None and pool_act is None and training:
        fused = conv_pool1d_forward(a, w_a, b_a, (bn_b, training))
    if fused is not None:
        p, pidx, k.c_shape = fused
        if keep:
            packs.append(None)
    elif pool_act is None:
        pooled = None
        if POOL_FUSION and ph == 2 and a_16 is not None and _l16_ok_for(a_16.shape, w_a, False):
            n_, _, h_, w_ = a_16.shape
            pf, pd = conv_l16_pack_pair(w_a, n_, h_, w_) if keep else (conv_l16_pack(w_a, n_, h_, w_, False), None)
            pooled = conv_l16_pool(a_16, w_a, b_a, prepacked=pf, stats_bn=(bn_b, training))     # conv + max-pool (+ bn_b statistics)
            if pooled is not None:
                p, pidx, k.c_shape = pooled
                if keep:
                    packs.append(pd)
            else:
                c = conv_l16(a_16, w_a, b_a, prepacked=pf)
                if keep:
                    packs.append(pd)
        else:
            c = _conv_fwd_any(a, a_16, w_a, b_a, a_max, packs)
        if pooled is None:
            p, pidx = maxpool_forward(c, ph)
            k.c_shape = tuple(c.shape)
            del c
    if pool_act is None:
        st_b = bn_prepare(p, bn_b, training, sync, counters, want_minmax=mm(p, w1), lazy=True)
        b, b_max, b_16 = _bn_fwd_for_conv(p, st_b, prelu_b.weight, w1, keep_f32=True, need_wgrad=keep)      # (the residual reads it)
        if (ph == 2 and not training and not keep and a_16 is not None and b_16 is not None
                and conv_l16_pool_act_supported(a_16.shape, w_a)):
            _act_calibrate(bn_b, w_a, b_16, b_a, prelu_b.weight)                 # (scaled limbs: the folded launch's scale next time)
    w2, b2 = _conv_params(res.conv2)
    w3, b3 = _conv_params(res.conv3)
    # inference: conv1 -> bn1 -> PReLU and conv2 -> bn2 -> PReLU as one launch each (the epilogue writes the next convolution's
    # limbs); training, or a layer the fold does not cover: convolution, statistics / range, apply pass
    fold = (not training) and not keep
    s1_16 = _eval_conv_act(b_16, w1, b1, res.bn1, res.prelu1.weight, w2) if fold else None
    if s1_16 is not None:
        r1 = s1 = st1 = None
        s1_max = s1_16.amax
    else:
        r1 = _conv_fwd_any(b, b_16, w1, b1, b_max, packs, (res.bn1, training))
        st1 = bn_prepare(r1, res.bn1, training, sync, counters, want_minmax=mm(r1, w2), lazy=True)
        s1, s1_max, s1_16 = _bn_fwd_for_conv(r1, st1, res.prelu1.weight, w2, need_wgrad=keep)
        if fold:
            _act_calibrate(res.bn1, w1, s1_16, b1, res.prelu1.weight)
    s2_16 = _eval_conv_act(s1_16, w2, b2, res.bn2, res.prelu2.weight, w3) if fold else None
    if s2_16 is not None:
        r2 = s2 = st2 = None
        s2_max = s2_16.amax
    else:
        r2 = _conv_fwd_any(s1, s1_16, w2, b2, s1_max, packs, (res.bn2, training))
        st2 = bn_prepare(r2, res.bn2, training, sync, counters, want_minmax=mm(r2, w3), lazy=True)
        s2, s2_max, s2_16 = _bn_fwd_for_conv(r2, st2, res.prelu2.weight, w3, need_wgrad=keep)
        if fold:
            _act_calibrate(res.bn2, w2, s2_16, b2, res.prelu2.weight)
    r3 = _conv_fwd_any(s2, s2_16, w3, b3, s2_max, packs, (res.bn3, training))
    st3 = bn_prepare(r3, res.bn3, training, sync, counters, lazy=True)
    feat, fidx = (None, None)
    next_stats = next_bn and (training or EVAL_L16)          # (inference: the next block's input BatchNorm wants the range)
    # small tensors (a channel is one workgroup): the output's statistics are not reduced here -- the next block's input BatchNorm
    # is such a unit itself (statistics + apply in one launch: cheaper than records, fold and finalisation launches)
    unit = bn_act_forward_unit(r3, st3, res.prelu3.weight, b, want_head) if st3.pending is not None else None
    if unit is not None:
        out, feat, fidx = unit
    elif FUSE_OUT_STATS and h_w_min(r3) * max(r3.shape[2], r3.shape[3]) > 1 and (want_head or next_stats):
        out, feat, fidx = bn_act_forward_rec(r3, st3, res.prelu3.weight, b, next_stats, want_head)
    else:
        out = bn_act_forward(r3, st3, res.prelu3.weight, residual=b)
        if want_head:
            feat, fidx = global_maxpool_forward(out)
    if keep:
        k.x, k.a, k.pidx, k.p, k.b = x, a, pidx, p, b
        k.r1, k.s1, k.r2, k.s2, k.r3 = r1, s1, r2, s2, r3
        k.st_a, k.st_b, k.st1, k.st2, k.st3 = st_a, st_b, st1, st2, st3
        k.fidx = fidx
        k.amax = (a_max, b_max, s1_max, s2_max)
        k.l16 = (a_16, b_16, s1_16, s2_16)
        k.packs = packs                    # [conv_a, conv1, conv2, conv3]
    if counters:
        if _COUNTER_SINK is not None:
            _COUNTER_SINK.extend(counters)
        else:
            torch._foreach_add_(counters, 1)
    return out, feat, k


def _block_params(mods):
    """Parameters in the order their gradients are returned by ConvBlockFn.backward."""
    bn_a, conv_a, _pool, bn_b, prelu_b, res = mods[0], mods[1], mods[2], mods[3], mods[4], mods[5]
    return [bn_a.weight, bn_a.bias, conv_a.weight, conv_a.bias, bn_b.weight, bn_b.bias, prelu_b.weight,
            res.conv1.weight, res.conv1.bias, res.bn1.weight, res.bn1.bias, res.prelu1.weight,
            res.conv2.weight, res.conv2.bias, res.bn2.weight, res.bn2.bias, res.prelu2.weight,
            res.conv3.weight, res.conv3.bias, res.bn3.weight, res.bn3.bias, res.prelu3.weight]


def stem_grads_pooled(x, st, dp, pidx, weight, chan_sum, bn):
    """Stem weight gradient straight from the pooled-resolution gradient `dp` and the pool indices, plus the parameter gradients
    of the BatchNorm `bn` in front of the stem, without the stem's input gradient (fsc_conv_stem_wgrad_pooled +
    fsc_conv_stem_grads_finish).  x = the BatchNorm's INPUT, st = its statistics: the kernel correlates the gradient
    with xhat = (x - mean) invstd, from which dW = gamma dW' + beta T, dgamma = sum w dW', dbeta = sum w T follow without any
    division by gamma.  chan_sum = per-channel total of dp.  Returns (dW, dgamma, dbeta), or None when the shape is not a stem
    layer."""
    n, c_in, h, w = x.shape
    c_out = weight.shape[0]
    d = _desc(n, c_in, c_out, h, w, 3, 3)
    blocks = _lib.load().fsc_conv_stem_wgrad_pooled_blocks(C.byref(d))
    if blocks == 0 or tuple(weight.shape[2:]) != (3, 3) or st.mean is None:
        return None
    part = torch.empty(blocks, c_out, 32, device=x.device, dtype=torch.float32)
    if TIMER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("fsc_conv_stem_wgrad_pooled", C.byref(d), ptr(x), ptr(st.mean), ptr(st.invstd), ptr(dp), ptr(pidx), ptr(part), stream_ptr())
    if TIMER is not None:
        e1.record()
        TIMER.records.append(("conv_stem_wgrad_pooled_kernel<%d>" % c_in, 2.0 * n * h * w * c_in * c_out * 9, e0, e1))
    dw = GRAD_OUT(weight) if GRAD_OUT is not None else None
    if dw is None:
        dw = torch.empty(tuple(weight.shape), device=x.device, dtype=torch.float32)
    dgamma, dbeta = _empty((c_in,), x), _empty((c_in,), x)
    call("fsc_conv_stem_grads_finish", C.byref(d), ptr(part), ptr(weight), ptr(chan_sum), ptr(bn.weight), ptr(bn.bias), 1,
         ptr(dw), None, ptr(dgamma), ptr(dbeta), stream_ptr())
    return dw, dgamma, dbeta


STEM_BN_IDENTITY = True


def h_w_min(t):
    return min(t.shape[2], t.shape[3])


GAMMA_FLOOR = 1e-3     # below this |gamma| the quotient in _stem_bn_grads amplifies fp32 cancellation: take the explicit route


class _GammaGuard:
    """min |gamma| of a BatchNorm weight, computed on the device when the block's forward is enqueued and read on the host
    when its backward is (a 4-byte pinned copy; the wait ends when the GPU has passed that point of the SAME step's forward,
    so the host still runs up to one step ahead)."""

    def __init__(self, gamma):
        dev = _empty((1,), gamma)
        call("fsc_absmin", ptr(gamma.detach()), gamma.numel(), ptr(dev), stream_ptr())
        self.host = torch.empty(1, dtype=torch.float32, pin_memory=True)
        self.host.copy_(dev, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self._keep = dev

    def ok(self):
        self.event.synchronize()
        v = float(self.host[0])
        return v == v and v >= GAMMA_FLOOR


def _stem_bn_grads(dc, dc_chan_sum, weight, dweight, bn, borders=None):
    """Parameter gradients (dgamma, dbeta) of the BatchNorm in FRONT of a 3x3 convolution, without the convolution's input
    gradient da -- for the first block, whose input (the log-mel image) needs no gradient, so da would only be summed:
        sum_p da[ci][p] a[ci][p] = sum_{co,tap} w[co][ci][tap] dW[co][ci][tap]            (a = the conv input = gamma xhat + beta)
        sum_p da[ci][p]          = sum_{co,tap} w[co][ci][tap] T[co][tap],   T[co][tap] = sum of dc[co] over the pixels whose
                                                                                           tap neighbour lies inside the image
    T follows from the channel totals of dc and its border sums (fsc_plane_border_sums).  dbeta = sum da,
    dgamma = sum da xhat = (sum da a - beta sum da) / gamma.  Saves the stem input-gradient kernel (1.36 ms at cfg 2) and the
    BN backward passes over the image; exact in exact arithmetic (same zero padding as the weight gradient)."""
    if borders is None:
        n, c, h, w = dc.shape
        b = torch.zeros(c, 8, device=dc.device, dtype=torch.float32)
        call("fsc_plane_border_sums", ptr(dc), n, c, h, w, ptr(b), stream_ptr())
    else:
        b, c = borders, borders.shape[0]           # (the un-pooled gradient itself was never materialised)
    s = dc_chan_sum
    row = [b[:, 0], None, b[:, 1]]           # rows excluded by ty = 0 (first row), 1 (none), 2 (last row)
    col = [b[:, 2], None, b[:, 3]]
    corner = {(0, 0): b[:, 4], (0, 2): b[:, 5], (2, 0): b[:, 6], (2, 2): b[:, 7]}
    t = torch.empty(c, 3, 3, device=b.device, dtype=torch.float32)
    for ty in range(3):
        for tx in range(3):
            v = s
            if row[ty] is not None:
                v = v - row[ty]
            if col[tx] is not None:
                v = v - col[tx]
            if (ty, tx) in corner:
                v = v + corner[(ty, tx)]
            t[:, ty, tx] = v
    dbeta = (weight * t[:, None]).sum((0, 2, 3))
    sa = (weight * dweight.reshape(weight.shape)).sum((0, 2, 3))
    gamma, beta = bn.weight.detach(), bn.bias.detach()
    dgamma = (sa - beta * dbeta) / gamma
    return dgamma, dbeta


FIRST_BLOCK_1D_IDENTITY = os.environ.get("FSC_FIRST_BLOCK_1D_IDENTITY", "1") == "1"


FIRST_BLOCK_1D_KERNEL = os.environ.get("FSC_FIRST_BLOCK_1D_KERNEL", "1") != "0"


def _first_block_grads_1d(x, st, dc, dc_chan_sum, weight, bn):
    """First block of the 1-d model (reference classifiers.py:147-154: BatchNorm1d -> Conv1d(k = 3) on the spectrogram, which needs no
    gradient): the convolution's weight gradient AND the BatchNorm's dgamma / dbeta from ONE weight-gradient pass over the RAW input x
    of the BatchNorm -- no input-gradient convolution (a 228 MB tensor at cfg 3 that was only ever summed), no BatchNorm backward
    pass over it, and NO DIVISION by gamma (the quotient form of _stem_bn_grads needs a host-side guard on min |gamma|, which a
    recorded HIP graph cannot evaluate).  With a = gamma xhat + beta inside the row and 0 in the padding, xhat = (x - mean) invstd:
        dWx[co][ci][t] = sum_p dc[co][p] x[ci][p + t - 1]        (x zero-extended: the same zero padding as the convolution)
        T[co][t]       = sum of dc[co] over the positions whose tap-t neighbour lies inside the row (total - first / last column)
        C = invstd (dWx - mean T) = sum dc xhat                   dW = gamma C + beta T
        dgamma[ci] = sum_p da xhat = sum_{co,t} w C               dbeta[ci] = sum_p da = sum_{co,t} w T
    (da = the input gradient nobody computes any more.)  Exact in exact arithmetic; in fp32 the difference dWx - mean T costs about
    log2(|mean| / std) bits of the correlation (log-magnitudes: one or two).  Returns (dW, dgamma, dbeta)."""
    c_out, c_in = weight.shape[0], weight.shape[1]
    dwx = conv_wgrad(x, dc, weight.shape, False)
    wgrad_flush(end=False)                                   # (read here)
    if FIRST_BLOCK_1D_KERNEL and bn.weight is not None and bn.bias is not None and c_out <= 2048:
        # the same arithmetic in one launch (fsc_first_block_1d_finish) instead of ~15 broadcast / reduction launches of torch
        out = GRAD_OUT(weight) if GRAD_OUT is not None else None
        dw = out if out is not None else torch.empty_like(weight)
        dgamma, dbeta = _empty((c_in,), x), _empty((c_in,), x)
        call("fsc_first_block_1d_finish", ptr(dwx), ptr(dc), ptr(dc_chan_sum), ptr(st.mean), ptr(st.invstd), ptr(bn.weight),
             ptr(bn.bias), ptr(weight), x.shape[0], c_in, c_out, dc.shape[3], ptr(dw), ptr(dgamma), ptr(dbeta), stream_ptr())
        return dw, dgamma, dbeta
    tot = dc_chan_sum
    t = torch.stack([tot - dc[:, :, 0, 0].sum(0), tot, tot - dc[:, :, 0, -1].sum(0)], 1)          # (c_out, 3)
    mu, istd = st.mean, st.invstd
    g, b = bn.weight.detach(), bn.bias.detach()
    core = (dwx.reshape(c_out, c_in, 3) - mu[None, :, None] * t[:, None, :]) * istd[None, :, None]
    w3 = weight.detach().reshape(c_out, c_in, 3)
    dgamma = (w3 * core).sum((0, 2))
    dbeta = (w3 * t[:, None, :]).sum((0, 2))
    dw = core * g[None, :, None] + b[None, :, None] * t[:, None, :]
    out = GRAD_OUT(weight) if GRAD_OUT is not None else None
    if out is not None:                                      # data-parallel: the gradient lives in the all-reduce bucket
        out.view(c_out, c_in, 3).copy_(dw)
        dw = out
    return dw.reshape(weight.shape), dgamma, dbeta


class ConvBlockFn(torch.autograd.Function):
    """One `conv_modules[k]` block of the reference (networks/classifiers.py:524-536 with
    ResnetBlock2d :72-104, or the 1-d pair :147-161 / :37-69) and its deep-supervision head
    (:589-591) as a single autograd node with a hand-written backward."""

    @staticmethod
    def forward(ctx, x, mods, training, want_head, ph, sync, next_bn, *params):
        keep = any(ctx.needs_input_grad)
        if keep and not training:
            raise _lib.FscError("conv_block: gradients in eval mode (BatchNorm on running statistics) are not on the "
                                "accelerated path; call model.train() or wrap the forward in torch.no_grad()")
        out, feat, k = _block_forward(x, mods, training, want_head, ph, keep, sync, next_bn)
        k.gamma_guard = None
        if (keep and STEM_BN_IDENTITY and not ctx.needs_input_grad[0] and mods[0].weight is not None and x.dim() == 4
                and tuple(_conv_params(mods[1])[0].shape[2:]) == (3, 3) and not (ph == 2 and _lib.load().fsc_conv_stem_wgrad_pooled_blocks(
                    C.byref(_desc(x.shape[0], x.shape[1], mods[1].weight.shape[0], x.shape[2], x.shape[3], 3, 3))))):
            k.gamma_guard = _GammaGuard(mods[0].weight)     # (the quotient route of _stem_bn_grads may be taken in backward)
        ctx.k = k
        ctx.sync = sync
        ctx.mods = mods
        ctx.ph = ph
        ctx.want_head = want_head
        ctx.x_needs_grad = ctx.needs_input_grad[0]
        ctx.set_materialize_grads(False)
        if want_head:
            return out, feat
        dummy = out.new_empty(0)
        ctx.mark_non_differentiable(dummy)
        return out, dummy

    @staticmethod
    def backward(ctx, d_out, d_feat):
        wgrad_begin()                       # the block's weight gradients reduce their split-K slices together (wgrad_flush)
        try:
            return ConvBlockFn._backward(ctx, d_out, d_feat)
        finally:
            wgrad_abort()                   # (nothing is left pending on the normal path)

    @staticmethod
    def _backward(ctx, d_out, d_feat):
        k, mods, ph, sync = ctx.k, ctx.mods, ctx.ph, ctx.sync
        if k is None:
            raise RuntimeError("ConvBlockFn: backward through the block a second time -- its saved activations are "
                               "freed after the first backward (retain_graph is not supported on this path)")
        bn_a, conv_a, _pool, bn_b, prelu_b, res = mods[0], mods[1], mods[2], mods[3], mods[4], mods[5]
        gmax = None
        if ctx.want_head and d_feat is not None:
            gmax = (d_feat.contiguous(), k.fidx)
        if d_out is not None:
            d_out = d_out.contiguous()
        if d_out is None and gmax is None:
            raise _lib.FscError("ConvBlockFn.backward: no upstream gradient")
        # ---- out = prelu3(bn3(r3) + b)
        a_max, b_max, s1_max, s2_max = k.amax
        w3, _ = _conv_params(res.conv3)
        w2, _ = _conv_params(res.conv2)
        w1, _ = _conv_params(res.conv1)
        wa, _ = _conv_params(conv_a)
        a_16, b_16, s1_16, s2_16 = k.l16
        pk_a, pk_1, pk_2, pk_3 = k.packs
        # gradients whose consumers (input-gradient and weight-gradient kernels) read L16 are written pre-split as well
        # (`*_m` is then that L16, else the amax buffer), and only as L16 when both consumers do
        w16, w32 = _grad_formats(k.r2.shape, w3)
        dr3, db, dg3, dbt3, dal3, dbias3, dr3_m = bn_act_backward(
            d_out, k.r3, k.st3, res.bn3, res.prelu3.weight, residual=k.b, gmax=gmax,
            want_dres=True, want_chan_sum=True, with_amax=True, sync=sync, l16=w16, want_f32=w32)
        s2_shape = k.r2.shape
        dw3 = _conv_wgrad_any(k.s2, s2_16, s2_max, dr3, _l16_of(dr3_m), _amax_of(dr3_m), w3)
        ds2 = _conv_dgrad_any(dr3, _l16_of(dr3_m), w3, s2_shape, _amax_of(dr3_m), prepacked=pk_3)
        del dr3, dr3_m
        w16, w32 = _grad_formats(k.r1.shape, w2)
        dr2, _, dg2, dbt2, dal2, dbias2, dr2_m = bn_act_backward(ds2, k.r2, k.st2, res.bn2, res.prelu2.weight,
                                                                 want_chan_sum=True, with_amax=True, sync=sync,
                                                                 l16=w16, want_f32=w32)
        del ds2
        dw2 = _conv_wgrad_any(k.s1, s1_16, s1_max, dr2, _l16_of(dr2_m), _amax_of(dr2_m), w2)
        ds1 = _conv_dgrad_any(dr2, _l16_of(dr2_m), w2, k.r1.shape, _amax_of(dr2_m), prepacked=pk_2)
        del dr2, dr2_m
        w16, w32 = _grad_formats(k.b.shape, w1)
        dr1, _, dg1, dbt1, dal1, dbias1, dr1_m = bn_act_backward(ds1, k.r1, k.st1, res.bn1, res.prelu1.weight,
                                                                 want_chan_sum=True, with_amax=True, sync=sync,
                                                                 l16=w16, want_f32=w32)
        del ds1
        dw1 = _conv_wgrad_any(k.b, b_16, b_max, dr1, _l16_of(dr1_m), _amax_of(dr1_m), w1)
        db = _conv_dgrad_any(dr1, _l16_of(dr1_m), w1, k.b.shape, _amax_of(dr1_m), accumulate_into=db, prepacked=pk_1)   # residual + conv1 paths
        del dr1, dr1_m
        # ---- b = prelu(bn_b(p))
        a_shape = tuple(k.c_shape[:1]) + (wa.shape[1],) + tuple(k.c_shape[2:])
        stem = None
        if (STEM_BN_IDENTITY and not ctx.x_needs_grad and ph == 2 and k.x.dim() == 4 and bn_a.weight is not None and k.st_a.mean is not None
                and _lib.load().fsc_conv_stem_wgrad_pooled_blocks(C.byref(_desc(a_shape[0], a_shape[1], wa.shape[0], a_shape[2], a_shape[3], 3, 3)))):
            # First block (its input needs no gradient): BN-b backward at the POOLED resolution, the stem weight gradient
            # straight from that and the pool indices, and bn_a's parameter gradients from the weight gradient --
            # the un-pooled gradient (2.8 GB at cfg 2), the stem input gradient and bn_a's backward passes never run.
            dp, _, dgb, dbtb, dalb, dbias_a = bn_act_backward(db.contiguous(), k.p, k.st_b, bn_b, prelu_b.weight,
                                                              want_chan_sum=True, sync=sync)
            del db
            stem = stem_grads_pooled(k.x, k.st_a, dp, k.pidx, wa, dbias_a, bn_a)
            dwa, dga, dbta = stem
            dx = None
            del dp
        if stem is None:
            w16, w32 = _grad_formats(a_shape, wa) if len(k.c_shape) == 4 else (False, True)
            dc, dgb, dbtb, dalb, dbias_a, dc_m = bn_act_backward_unpool(db.contiguous(), k.p, k.st_b, bn_b, prelu_b.weight,
                                                                        k.pidx, k.c_shape, ph, sync=sync, l16=w16, want_f32=w32)
            del db
            first_1d = (FIRST_BLOCK_1D_IDENTITY and not ctx.x_needs_grad and dc is not None and dc.dim() == 4 and dc.shape[2] == 1
                        and tuple(wa.shape[2:]) == (1, 3) and dc.shape[3] >= 2 and bn_a.weight is not None and k.st_a.mean is not None
                        and sync is None)
            if first_1d:
                # first block of the 1-d model: everything from one weight-gradient pass over bn_a's raw input (no dgrad, no BN backward)
                dwa, dga, dbta = _first_block_grads_1d(k.x, k.st_a, dc, dbias_a, wa, bn_a)
                dx = None
                del dc, dc_m
            else:
                dwa = _conv_wgrad_any(k.a, a_16, a_max, dc, _l16_of(dc_m), _amax_of(dc_m), wa)
            if first_1d:
                pass
            elif (STEM_BN_IDENTITY and not ctx.x_needs_grad and dc is not None and dc.dim() == 4 and tuple(wa.shape[2:]) == (3, 3)
                    and bn_a.weight is not None and h_w_min(dc) >= 2 and k.gamma_guard is not None and k.gamma_guard.ok()):
                # the block input needs no gradient: bn_a's parameter gradients from the weight gradient (no dgrad, no BN backward)
                wgrad_flush(end=False)                  # (dwa is read here)
                if L16_WGRAD_SIDE == 2:
                    l16_wgrad_fork(dc.device)
                    l16_wgrad_join(dc.device)
                dga, dbta = _stem_bn_grads(dc, dbias_a, wa, dwa, bn_a)
                dx = None
                del dc, dc_m
            else:
                da = _conv_dgrad_any(dc, _l16_of(dc_m), wa, a_shape, _amax_of(dc_m), prepacked=pk_a)
                del dc, dc_m
                dx, _, dga, dbta, _, _ = bn_act_backward(da, k.x, k.st_a, bn_a, sync=sync, want_dx=ctx.x_needs_grad)

        def like(param, g):
            return g.reshape(param.shape) if g is not None else None

        wgrad_flush()
        if L16_WGRAD_SIDE == 2:
            l16_wgrad_fork(k.x.device)
            _WGRAD.l16_inflight = False
        join_side_stream(k.x.device)           # weight gradients computed on the side stream
        grads = [dga, dbta, like(conv_a.weight, dwa), dbias_a, dgb, dbtb, dalb,
                 like(res.conv1.weight, dw1), dbias1, dg1, dbt1, dal1,
                 like(res.conv2.weight, dw2), dbias2, dg2, dbt2, dal2,
                 like(res.conv3.weight, dw3), dbias3, dg3, dbt3, dal3]
        ctx.k = None
        return (dx, None, None, None, None, None, None) + tuple(grads)


def conv_block(x, mods, training, want_head, ph, sync=None, next_bn=False):
    """Differentiable block call.  Returns (out, feat or None).  `sync`: cross-replica BN statistics (bn_prepare).
    next_bn: the output goes straight into another block (whose input BatchNorm then finds its statistics reduced)."""
    _need_cuda(x, "conv_block")
    x = x.contiguous()
    if not training:
        sync = None
    if torch.is_grad_enabled() and any(p.requires_grad for p in _block_params(mods)):
        out, feat = ConvBlockFn.apply(x, mods, training, want_head, ph, sync, next_bn, *_block_params(mods))
    else:
        out, feat, _ = _block_forward(x, mods, training, want_head, ph, keep=False, sync=sync, next_bn=next_bn)
    return out, (feat if want_head else None)


# ------------------------------------------------------------------------------ head ops
class BNActFn(torch.autograd.Function):
    """BatchNorm1d (+ PReLU) on (N, C) features (classifiers.py:543-546)."""

    @staticmethod
    def forward(ctx, x, bn, prelu, training, sync, gamma, beta, alpha):
        x = x.contiguous()
        st = bn_prepare(x, bn, training, sync if training else None, lazy=True)
        y = bn_act_forward(x, st, alpha)
        ctx.save_for_backward(x)
        ctx.st, ctx.bn, ctx.prelu, ctx.sync = st, bn, prelu, (sync if training else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        st = ctx.st
        if st.mean is None:
            raise _lib.FscError("BNActFn.backward in eval mode is not supported")
        alpha = ctx.prelu.weight if ctx.prelu is not None else None
        dx, _, dg, db, dal, _ = bn_act_backward(dy.contiguous(), x, st, ctx.bn, alpha, sync=ctx.sync)
        return dx, None, None, None, None, dg, db, dal


_COUNTER_SINK = None       # num_batches_tracked tensors of the BatchNorms a training forward has gone through so far


def counters_begin():
    """From here on the training-mode BatchNorms only note their num_batches_tracked; counters_flush() adds 1 to all of them
    in one launch (torch/nn/modules/batchnorm.py bumps each in its own forward: same values once the forward is over)."""
    global _COUNTER_SINK
    _COUNTER_SINK = []


def counters_flush():
    global _COUNTER_SINK
    sink, _COUNTER_SINK = _COUNTER_SINK, None
    if sink:
        table = (C.c_void_p * len(sink))(*[ptr(t) for t in sink])
        call("fsc_bump_counters", table, len(sink), stream_ptr())


def _cat_cols(pieces, widths, rows, full, split):
    table = (C.c_void_p * len(pieces))(*[ptr(t) for t in pieces])
    wtab = (C.c_int * len(pieces))(*widths)
    call("fsc_cat_cols", table, wtab, len(pieces), rows, ptr(full), split, stream_ptr())


class CatColsFn(torch.autograd.Function):
    """torch.cat(pieces, -1) of 2-d fp32 tensors (classifiers.py:595: the deep-supervision features) and its way back, one
    launch each."""

    @staticmethod
    def forward(ctx, *pieces):
        pieces = [t.contiguous() for t in pieces]
        rows, widths = pieces[0].shape[0], [int(t.shape[1]) for t in pieces]
        out = _empty((rows, sum(widths)), pieces[0])
        _cat_cols(pieces, widths, rows, out, 0)
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        outs = [_empty((d.shape[0], w), d) for w in ctx.widths]
        _cat_cols(outs, ctx.widths, d.shape[0], d, 1)
        return tuple(outs)


def cat_features(feats):
    if len(feats) == 1:
        return feats[0]
    if len(feats) > 16 or any(t.dim() != 2 or t.dtype != torch.float32 for t in feats):
        return torch.cat(feats, -1)
    return CatColsFn.apply(*feats)


def bn_act(x, bn, prelu, training, sync=None):
    alpha = prelu.weight if prelu is not None else None
    return BNActFn.apply(x, bn, prelu, training, sync, bn.weight, bn.bias, alpha)


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        m, k = x.shape
        n_out = weight.shape[0]
        y = _empty((m, n_out), x)
        call("fsc_linear_fwd", ptr(x), ptr(weight), ptr(bias), ptr(y), m, k, n_out, stream_ptr())
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        m, k = x.shape
        n_out = weight.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        db = _empty((n_out,), x) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        call("fsc_linear_bwd", ptr(dy), ptr(x), ptr(weight), ptr(dx), ptr(dw), ptr(db), m, k, n_out,
             stream_ptr())
        return dx, dw, db


def linear(x, weight, bias):
    return LinearFn.apply(x, weight, bias)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        x = x.contiguous()
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        call("fsc_dropout_fwd", ptr(x), ptr(y), ptr(mask), x.numel(), p, seed, offset, stream_ptr())
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call("fsc_dropout_bwd", ptr(dy), ptr(mask), ptr(dx), dy.numel(), ctx.p, stream_ptr())
        return dx, None, None, None


class DropoutState:
    """Position in the counter-based random stream of one dropout site (owned by the caller, e.g. the model:
    neither the library nor this module keeps it)."""

    def __init__(self):
        self.offset = 0


def dropout(x, p, training, state=None):
    """nn.Dropout semantics (classifiers.py:547).  `state` advances by x.numel() per call; without one every call
    draws from the start of the stream of torch.initial_seed()."""
    if not training or p <= 0.0:
        return x
    offset = 0
    if state is not None:
        offset = state.offset
        state.offset += x.numel()
    return DropoutFn.apply(x, float(p), int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, offset)


# ------------------------------------------------------------------------------ losses
class LsepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        logits = logits.contiguous()
        targets = targets.contiguous().float()
        n, c = logits.shape
        loss = _empty((n,), logits)
        call("fsc_lsep_fwd", ptr(logits), ptr(targets), ptr(loss), n, c, stream_ptr())
        ctx.save_for_backward(logits, targets)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, targets = ctx.saved_tensors
        n, c = logits.shape
        dl = torch.empty_like(logits)
        call("fsc_lsep_bwd", ptr(logits), ptr(targets), ptr(dloss.contiguous()), ptr(dl), n, c, stream_ptr())
        return dl, None


class MeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = x.contiguous()
        y = _empty((), x)
        call("fsc_mean_fwd", ptr(x), ptr(y), x.numel(), scale, stream_ptr())
        ctx.shape, ctx.scale = x.shape, scale
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        call("fsc_mean_bwd", ptr(dy.contiguous()), ptr(dx), dx.numel(), ctx.scale, stream_ptr())
        return dx, None


def mean(x, scale=1.0):
    return MeanFn.apply(x, float(scale))


class BceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        logits = logits.contiguous()
        targets = targets.contiguous().float()
        loss = _empty((), logits)
        ws = torch.empty(1, device=logits.device, dtype=torch.float64)
        call("fsc_bce_fwd", ptr(logits), ptr(targets), ptr(loss), ptr(ws), logits.numel(), stream_ptr())
        ctx.save_for_backward(logits, targets)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, targets = ctx.saved_tensors
        dl = torch.empty_like(logits)
        call("fsc_bce_bwd", ptr(logits), ptr(targets), ptr(dloss.contiguous()), ptr(dl), logits.numel(),
             stream_ptr())
        return dl, None


def sigmoid(x):
    _need_cuda(x, "sigmoid")
    x = x.detach().contiguous()
    y = torch.empty_like(x)
    call("fsc_sigmoid", ptr(x), ptr(y), x.numel(), stream_ptr())
    return y


# ------------------------------------------------------------------------------ MixUp on device
def mixup_rows(a, b, partner, len_a, len_b, start, alpha, labels_a=None, labels_b=None):
    """mixup_batch with a partner table (fsc_mixup_rows): row n of `a` (N, Ta) is mixed with row partner[n] of `b` (M, Tb), whose
    true length is len_b[n]; partner[n] < 0 passes row n (and its labels) through unchanged -- MixUp.p not drawn,
    ops/transforms.py:57 -- without reading `b`.  No gathered copy of the partners.  Returns (mixed, labels)."""
    _need_cuda(a, "mixup_rows")
    a, b = a.contiguous(), b.contiguous()
    n, ta = a.shape
    tb = b.shape[1]
    t_out = max(ta, tb)
    out = _empty((n, t_out), a)
    alpha64 = np.asarray(alpha, dtype=np.float64)
    al = torch.from_numpy(alpha64.astype(np.float32)).to(a.device)
    om = torch.from_numpy((1.0 - alpha64).astype(np.float32)).to(a.device)
    dev = lambda v: torch.as_tensor(np.asarray(v, dtype=np.int32)).to(a.device)  # noqa: E731
    pr, la, lb, st = dev(partner), dev(len_a), dev(len_b), dev(start)
    lo = None
    c = 0
    if labels_a is not None:
        labels_a, labels_b = labels_a.contiguous().float(), labels_b.contiguous().float()
        lo = torch.empty_like(labels_a)
        c = labels_a.shape[1]
    call("fsc_mixup_rows", ptr(a), ptr(b), ptr(pr), ptr(la), ptr(lb), ptr(st), ptr(al), ptr(om), ptr(out), n, ta, tb, t_out,
         ptr(labels_a), ptr(labels_b), ptr(lo), c, stream_ptr())
    return out, lo


def mixup_batch(a, b, len_a, len_b, start, alpha, labels_a=None, labels_b=None):
    """Batched ops/audio.py:32-52.  a (N, Ta), b (N, Tb) zero-padded rows with true lengths
    len_a / len_b (int32); `alpha` is the fp64 mixing draw per row.  Returns (mixed, labels)."""
    _need_cuda(a, "mixup_batch")
    n, ta = a.shape
    tb = b.shape[1]
    t_out = max(ta, tb)
    out = _empty((n, t_out), a)
    alpha64 = np.asarray(alpha, dtype=np.float64)
    al = torch.from_numpy(alpha64.astype(np.float32)).to(a.device)
    om = torch.from_numpy((1.0 - alpha64).astype(np.float32)).to(a.device)
    dev = lambda v: torch.as_tensor(np.asarray(v, dtype=np.int32)).to(a.device)  # noqa: E731
    la, lb, st = dev(len_a), dev(len_b), dev(start)
    lo = None
    c = 0
    if labels_a is not None:
        labels_a, labels_b = labels_a.contiguous().float(), labels_b.contiguous().float()
        lo = torch.empty_like(labels_a)
        c = labels_a.shape[1]
    call("fsc_mixup_batch", ptr(a.contiguous()), ptr(b.contiguous()), ptr(la), ptr(lb), ptr(st), ptr(al),
         ptr(om), ptr(out), n, ta, tb, t_out, ptr(labels_a), ptr(labels_b), ptr(lo), c, stream_ptr())
    return out, lo
