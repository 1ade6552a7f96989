"""Tensor-level wrappers over the C ABI.  Tensors are fp32 HIP tensors, activations laid out
[B][H][W][C] (contiguous), conv weights logical OIHW stored channels_last (= [Cout][kh][kw][Cin]).
Torch is used for allocation and the stream handle only; every arithmetic op is a HIP kernel."""
import ctypes as C

import torch

from . import lib as L

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
PREC_F32, PREC_BF16, PREC_F32_WINO4 = 0, 1, 2
_precision = PREC_F32


def set_precision(name):
    """'f32' (default) or 'bf16': arithmetic of the conv inner products (bf16 operands, fp32 accumulate; everything in
    memory stays fp32).  Process-wide switch used by bench.py / train(config) for BASELINE config 3."""
    global _precision
    _precision = {"f32": PREC_F32, "fp32": PREC_F32, "bf16": PREC_BF16}[name]


def get_precision():
    return "bf16" if _precision == PREC_BF16 else "f32"
BN_EPS, BN_MOMENTUM = 1e-5, 0.1
PROF_TAGS = False        # bench.py's instrumented pass: label conv launches with their layer (tg_prof_tag)


def tag(name):
    if PROF_TAGS:
        _lib().tg_prof_tag(name.encode())

_ws = {}


def _lib():
    return L.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise L.TgError(f"{name}: expected a HIP (cuda) tensor -- there is no CPU path")
    if t.dtype != torch.float32:
        raise L.TgError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise L.TgError(f"{name}: expected a contiguous tensor, strides {t.stride()}")


def workspace(nbytes):
    """Stream-ordered scratch shared by all ops on (device, stream); grown on demand."""
    dev = torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream().cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        n = max(int(nbytes) // 4 + 64, 1 << 20)
        buf = torch.empty(n, dtype=torch.float32, device=f"cuda:{dev}")
        _ws[key] = buf
    return buf


def empty(*shape, like=None, device=None):
    return torch.empty(*shape, dtype=torch.float32, device=like.device if like is not None else device)


def weight_view(w):
    """[Cout][kh][kw][Cin] view of an OIHW parameter stored channels_last; converts the storage
    in place (once) if the parameter is not laid out that way yet."""
    v = w.detach().permute(0, 2, 3, 1)
    if not v.is_contiguous():
        w.data = w.data.contiguous(memory_format=torch.channels_last)
        v = w.detach().permute(0, 2, 3, 1)
        if not v.is_contiguous():          # ambiguous strides (Cin == 1 or k == 1): force a dense OHWI buffer
            dense = v.contiguous()
            w.data = dense.permute(0, 3, 1, 2)
            v = w.detach().permute(0, 2, 3, 1)
    return v


def _prec(wino4):
    """TG_PREC_F32_WINO4 (Winograd F(4x4,3x3) where the geometry allows: the frozen VGG trunk) only refines fp32 mode."""
    return PREC_F32_WINO4 if (wino4 and _precision == PREC_F32) else _precision


def conv_geom(x, cout, k, stride, pad, wino4=False):
    B, H, W, Cin = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return L.TgConv(B, H, W, Cin, Ho, Wo, cout, k, stride, pad, _prec(wino4))


# ---- prepared weights (tg_conv_wprep): computed when a weight tensor is first used and again only after it changed --------
WPREP_FWD, WPREP_DGRAD = 0, 1
WPREP_CACHE = True            # False: every conv call prepares its weights in the workspace (the round-1 behaviour)
WPREP_VERIFY = __import__("os").environ.get("TG_WPREP_VERIFY") == "1"      # debug: re-prepare and compare on every cache hit
WPREP_BATCH = True            # re-prepare all batchable entries of the updated weights in ONE launch right after the optimiser step
_wprep = {}                   # (weight ptr, mode, geometry) -> [weakref(weight), version stamp, prepared buffer or None,
                              #                                   batch descriptor (bytes) or False, used since the last batch]
_wstamp = {}                  # weight ptr -> number of out-of-band updates (kernels writing through raw pointers)
_wprep_tables = {}            # (device, concatenated descriptor bytes) -> device array of those descriptors
wprep_capture_log = None      # tg_hip.graph, while a step is captured: the keys whose batched preparation the graph replays


def weights_updated(params):
    """REQUIRED after any parameter write torch's version counter does not see -- kernels writing through raw pointers
    (hip_adam_step does it itself), `.data` writes, custom broadcasts, user-side EMA / clamping through `.data`: the
    prepared-weight cache keeps its own stamp per storage and would otherwise keep convolving with stale transforms.
    TG_WPREP_VERIFY=1 (debug) re-prepares on every use and raises when a cached buffer was stale."""
    ptrs = set()
    for p in params:
        if p.dim() == 4:
            ptr = p.data_ptr()
            _wstamp[ptr] = _wstamp.get(ptr, 0) + 1
            ptrs.add(ptr)
    if WPREP_BATCH and WPREP_CACHE and ptrs:
        _prepare_batch(ptrs)


def _prepare_batch(ptrs):
    """The prepared forms of the weights at `ptrs` that were used since the last batch, recomputed by ONE launch
    (tg_conv_wprep_run over a cached device table of descriptors) instead of one launch per layer and mode at first use."""
    ents = []
    for key, e in _wprep.items():
        if key[0] in ptrs and e[3] and e[4] and e[0]() is not None:
            ents.append((key, e))
    if not ents:
        return
    # keyed by the descriptors themselves (they hold the w / out pointers, mode and geometry): an id()-based key could be
    # recycled by CPython for a replaced entry and hit a stale table whose `out` points at a freed prepared buffer
    raw = b"".join(e[3] for _, e in ents)
    tkey = (ents[0][1][2].device, raw)
    table = _wprep_tables.get(tkey)
    if table is None:
        import numpy as np
        table = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(ents[0][1][2].device)
        if len(_wprep_tables) > 16:
            _wprep_tables.clear()
        _wprep_tables[tkey] = table
    L.check(_lib().tg_conv_wprep_run(C.c_void_p(table.data_ptr()), len(ents), _stream()), "tg_conv_wprep_run")
    for key, e in ents:
        e[1] = (e[0]()._version, _wstamp.get(key[0], 0))
        e[4] = False
    if wprep_capture_log is not None:
        wprep_capture_log.extend(key for key, _e in ents)


def graph_replayed(weight_ptrs, refreshed_keys):
    """A captured train step was replayed: the weights at `weight_ptrs` changed on the device behind the host's back.  Every
    prepared form of them is stale from here on -- entries made eagerly for another geometry (validation / inference at
    another size), and the non-batchable ones the graph re-prepares in its FORWARD, i.e. from the weights of before its
    Adam -- except those in `refreshed_keys`, which the replayed tg_conv_wprep_run launch has just recomputed."""
    for ptr in weight_ptrs:
        _wstamp[ptr] = _wstamp.get(ptr, 0) + 1
    for key in refreshed_keys:
        e = _wprep.get(key)
        if e is not None:
            w = e[0]()
            if w is not None:
                e[1] = (w._version, _wstamp.get(key[0], 0))


def _prepared(w, wv, g, mode):
    """Prepared form of weight `w` for (g, mode), or None when that path reads the raw weights."""
    if not WPREP_CACHE:
        return None
    import weakref
    ptr = wv.data_ptr()
    key = (ptr, mode, g.H, g.W, g.Cin, g.Cout, g.k, g.stride, g.pad, g.precision)
    stamp = (w._version, _wstamp.get(ptr, 0))
    ent = _wprep.get(key)
    if ent is not None and ent[0]() is w and ent[1] == stamp:
        ent[4] = True
        if WPREP_VERIFY and ent[2] is not None:
            fresh = torch.empty_like(ent[2])
            L.check(_lib().tg_conv_wprep(C.byref(g), mode, _p(wv), _p(fresh), _stream()), "tg_conv_wprep")
            if not torch.equal(fresh.view(torch.int32), ent[2].view(torch.int32)):
                raise L.TgError("prepared weights are stale: a parameter was rewritten without tg_hip.ops.weights_updated()")
        return ent[2]
    lib = _lib()
    if ent is None or ent[0]() is not w:            # first use (or the address was recycled for another tensor)
        nb = lib.tg_conv_wprep_bytes(C.byref(g), mode)
        buf = torch.empty((nb + 3) // 4, dtype=torch.float32, device=wv.device) if nb else None
        if len(_wprep) > 512:
            for k_ in [k_ for k_, e in _wprep.items() if e[0]() is None]:
                del _wprep[k_]
        item = False
        if buf is not None:
            raw = C.create_string_buffer(lib.tg_conv_wprep_item_bytes())
            if lib.tg_conv_wprep_item(C.byref(g), mode, _p(wv), _p(buf), raw):
                item = raw.raw
        ent = [weakref.ref(w), None, buf, item, True]
        _wprep[key] = ent
    if ent[2] is not None:
        L.check(lib.tg_conv_wprep(C.byref(g), mode, _p(wv), _p(ent[2]), _stream()), "tg_conv_wprep")
    ent[1] = stamp
    ent[4] = True
    return ent[2]


def conv_fwd(x, w, bias, k, stride, pad, in_mask=None, ratio=None, act=ACT_NONE, slope=0.0, wino4=False, pool=False):
    """pool=True: returns (y, maxpool2(y)) from one call (tg_conv_fwd_pool: the pooled tensor leaves the conv's output transform
    where the kernel allows, the pool kernel runs on y otherwise)."""
    _chk(x, "x"); _chk(bias, "bias"); _chk(in_mask, "in_mask"); _chk(ratio, "ratio")
    wv = weight_view(w)
    _chk(wv, "weight")
    g = conv_geom(x, wv.shape[0], k, stride, pad, wino4)
    assert wv.shape == (g.Cout, k, k, g.Cin), (tuple(wv.shape), g.Cout, k, g.Cin)
    y = empty(g.B, g.Ho, g.Wo, g.Cout, like=x)
    lib = _lib()
    nb = lib.tg_conv_fwd_ws_bytes(C.byref(g))
    ws = workspace(nb)
    if pool:
        yp = empty(g.B, g.Ho // 2, g.Wo // 2, g.Cout, like=x)
        L.check(lib.tg_conv_fwd_pool(C.byref(g), _p(x), _p(in_mask), _p(wv), _p(_prepared(w, wv, g, WPREP_FWD)), _p(bias), _p(ratio),
                                     act, slope, _p(y), _p(yp), _p(ws), ws.numel() * 4, _stream()), "tg_conv_fwd_pool")
        return y, yp
    L.check(lib.tg_conv_fwd_p(C.byref(g), _p(x), _p(in_mask), _p(wv), _p(_prepared(w, wv, g, WPREP_FWD)), _p(bias), _p(ratio),
                              act, slope, _p(y), _p(ws), ws.numel() * 4, _stream()), "tg_conv_fwd")
    return y


def _bn_act(in_bn):
    """(mean, rstd, gamma, beta[, act[, slope]]) -> TgBnAct (the tensors stay referenced by the caller's tuple)."""
    mean, rstd, gamma, beta = in_bn[:4]
    for t in (mean, rstd, gamma, beta):
        _chk(t, "in_bn")
    act = in_bn[4] if len(in_bn) > 4 else ACT_RELU
    slope = in_bn[5] if len(in_bn) > 5 else 0.0
    return L.TgBnAct(mean.data_ptr(), rstd.data_ptr(), gamma.detach().data_ptr(), beta.detach().data_ptr(), act, slope)


def conv_bnin_supported(x_shape, cout, k, stride, pad, wgrad=False):
    """Can conv_fwd_bnin / conv_wgrad_bnin take this layer (input = act(BN(x)) applied while the kernel stages x)?"""
    B, H, W, Cin = x_shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    g = L.TgConv(B, H, W, Cin, Ho, Wo, cout, k, stride, pad, _precision)
    return bool(_lib().tg_conv_bnin_supported(C.byref(g), 1 if wgrad else 0))


def conv_fwd_bnin(x, in_bn, w, bias, k, stride, pad, act=ACT_NONE, slope=0.0):
    """conv(act_bn(BN(x)), w) + bias with the BatchNorm + activation of `in_bn` applied on load (tg_conv_fwd_bnin)."""
    _chk(x, "x"); _chk(bias, "bias")
    wv = weight_view(w)
    _chk(wv, "weight")
    g = conv_geom(x, wv.shape[0], k, stride, pad)
    y = empty(g.B, g.Ho, g.Wo, g.Cout, like=x)
    lib = _lib()
    ws = workspace(lib.tg_conv_fwd_ws_bytes(C.byref(g)))
    bn = _bn_act(in_bn)
    L.check(lib.tg_conv_fwd_bnin(C.byref(g), _p(x), C.byref(bn), _p(wv), _p(bias), act, slope, _p(y), _p(ws), ws.numel() * 4, _stream()),
            "tg_conv_fwd_bnin")
    return y


def conv_pool_code_supported(x_shape, cout):
    """Can conv_fwd_pool_code take this 3x3 / stride-1 / pad-1 layer (pooled tensor + code, conv output never written)?"""
    B, H, W, Cin = x_shape
    g = L.TgConv(B, H, W, Cin, H, W, cout, 3, 1, 1, _precision)
    return bool(_lib().tg_conv_pool_code_supported(C.byref(g)))


def conv_fwd_pool_code(x, w, bias):
    """3x3 / stride-1 / pad-1 conv -> ReLU -> 2x2 max-pool: returns (pooled, code); the full-resolution output is not written
    (tg_conv_fwd_pool_code).  code: uint8 [B][H/2][W/2][Cout], consumed by maxpool2_bwd_code."""
    _chk(x, "x"); _chk(bias, "bias")
    wv = weight_view(w)
    _chk(wv, "weight")
    g = conv_geom(x, wv.shape[0], 3, 1, 1)
    yp = empty(g.B, g.Ho // 2, g.Wo // 2, g.Cout, like=x)
    code = torch.empty((g.B, g.Ho // 2, g.Wo // 2, g.Cout), dtype=torch.uint8, device=x.device)
    lib = _lib()
    ws = workspace(lib.tg_conv_fwd_ws_bytes(C.byref(g)))
    L.check(lib.tg_conv_fwd_pool_code(C.byref(g), _p(x), _p(wv), _p(_prepared(w, wv, g, WPREP_FWD)), _p(bias), _p(yp),
                                      C.c_void_p(code.data_ptr()), _p(ws), ws.numel() * 4, _stream()), "tg_conv_fwd_pool_code")
    return yp, code


def maxpool2_bwd_code(dout, code):
    """Gradient in front of the fused conv's ReLU from the pooled gradient and the pool code: [B][2 Ho][2 Wo][C]."""
    _chk(dout, "dout")
    B, Ho, Wo, Cc = dout.shape
    assert code.dtype == torch.uint8 and code.is_contiguous() and tuple(code.shape[1:]) == (Ho, Wo, Cc) and code.shape[0] >= B
    dx = empty(B, 2 * Ho, 2 * Wo, Cc, like=dout)
    L.check(_lib().tg_maxpool2_bwd_code(_p(dout), C.c_void_p(code.data_ptr()), B, Ho, Wo, Cc, _p(dx), _stream()), "tg_maxpool2_bwd_code")
    return dx


def conv_dgrad(dy, w, x_shape, k, stride, pad, in_mask=None, out=None, gate=None, gate_act=ACT_RELU, gate_slope=0.0, wino4=False):
    """dx for an input of shape x_shape=[B,H,W,Cin]; accumulates into `out` when given.  `gate` = output of the
    activation that produced x: its backward is fused into the epilogue (dx *= act'(gate))."""
    _chk(dy, "dy"); _chk(in_mask, "in_mask"); _chk(out, "out"); _chk(gate, "gate")
    wv = weight_view(w)
    B, H, W, Cin = x_shape
    g = L.TgConv(B, H, W, Cin, dy.shape[1], dy.shape[2], dy.shape[3], k, stride, pad, _prec(wino4))
    acc = 1 if out is not None else 0
    dx = out if out is not None else empty(B, H, W, Cin, like=dy)
    lib = _lib()
    ws = workspace(lib.tg_conv_dgrad_ws_bytes(C.byref(g)))
    if gate is not None:
        assert out is None and tuple(gate.shape) == tuple(x_shape)
    L.check(lib.tg_conv_dgrad_p(C.byref(g), _p(dy), _p(wv), _p(_prepared(w, wv, g, WPREP_DGRAD)), _p(in_mask), _p(gate),
                                gate_act if gate is not None else ACT_NONE, gate_slope, _p(dx), acc, _p(ws), ws.numel() * 4,
                                _stream()), "tg_conv_dgrad")
    return dx


def conv_wgrad(x, dy, w, k, stride, pad, in_mask=None, want_bias=True, dw_out=None, db_out=None, in_bn=None):
    """Returns (dw, db): dw has the parameter's logical shape AND strides (channels_last).  dw_out / db_out:
    preallocated destinations (persistent gradient buffers) with the parameter's layout.
    in_bn: the layer's input is act(BN(x)), applied on load (tg_conv_wgrad_bnin; see conv_bnin_supported)."""
    _chk(x, "x"); _chk(dy, "dy"); _chk(in_mask, "in_mask")
    wv = weight_view(w)
    B, H, W, Cin = x.shape
    g = L.TgConv(B, H, W, Cin, dy.shape[1], dy.shape[2], dy.shape[3], k, stride, pad, _precision)
    if dw_out is not None:
        dwv = dw_out.permute(0, 2, 3, 1)
        assert dwv.is_contiguous() and dwv.shape == wv.shape
    else:
        dwv = torch.empty_like(wv)                   # [Cout][k][k][Cin] contiguous
    db = (db_out if db_out is not None else empty(g.Cout, like=x)) if want_bias else None
    lib = _lib()
    ws = workspace(lib.tg_conv_wgrad_ws_bytes(C.byref(g)))
    if in_bn is not None:
        assert in_mask is None
        bn = _bn_act(in_bn)
        L.check(lib.tg_conv_wgrad_bnin(C.byref(g), _p(x), C.byref(bn), _p(dy), _p(dwv), _p(db), _p(ws), ws.numel() * 4, _stream()),
                "tg_conv_wgrad_bnin")
        return dwv.permute(0, 3, 1, 2), db
    L.check(lib.tg_conv_wgrad(C.byref(g), _p(x), _p(in_mask), _p(dy), _p(dwv), _p(db), _p(ws), ws.numel() * 4, _stream()),
            "tg_conv_wgrad")
    return dwv.permute(0, 3, 1, 2), db


def fold_cin(w):
    """[Cout,Cin,k,k] -> [Cout,1,k,k] channel-summed kernel (grey image repeated xCin)."""
    wv = weight_view(w)
    co, kh, kw, ci = wv.shape
    out = empty(co, kh, kw, 1, like=wv)
    L.check(_lib().tg_fold_cin(_p(wv), co, kh * kw, ci, _p(out), _stream()), "tg_fold_cin")
    return out.permute(0, 3, 1, 2)


def mask_update(mask, k, stride, pad):
    _chk(mask, "mask")
    B, H, W = mask.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    mo, ratio = empty(B, Ho, Wo, like=mask), empty(B, Ho, Wo, like=mask)
    L.check(_lib().tg_mask_update(_p(mask), B, H, W, k, stride, pad, Ho, Wo, _p(mo), _p(ratio), _stream()), "tg_mask_update")
    return mo, ratio


def mask_up_merge(up_mask, skip_mask):
    _chk(up_mask, "up_mask"); _chk(skip_mask, "skip_mask")
    B, h, w = up_mask.shape
    _, H, W = skip_mask.shape
    out = torch.empty_like(skip_mask)
    L.check(_lib().tg_mask_up_merge(_p(up_mask), _p(skip_mask), B, h, w, H, W, _p(out), _stream()), "tg_mask_up_merge")
    return out


MASK_FUSE_PIXELS = 64 * 64


def mask_pyramid(mask, enc, dec):
    """Every mask of one generator forward from ONE launch (tg_mask_pyramid).  enc / dec: [(k, stride, pad), ...] of the
    encoder / decoder partial convs (generator.py:13-28).  Returns (m, er, dmasks, dr) exactly as the per-level calls
    would: m[i] / er[i] = mask / ratio after encoder layer i (m[0] = input, er[0] = None), dmasks[j] = merged mask fed to
    decoder layer j (generator.py:51-54,68-74), dr[j] = its ratio map.  Bit-identical to mask_update / mask_up_merge."""
    _chk(mask, "mask")
    B, H, W = mask.shape
    ne, nd = len(enc), len(dec)
    dims = [(H, W)]
    for (k, s, p) in enc:
        h, w = dims[-1]
        dims.append(((h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1))
    # decoder level j merges the current decoder mask (upsampled x2) with skip mask m[ne-1-j] (the input mask for the last)
    sizes = [B * h * w for (h, w) in dims[1:]] * 2
    ddims = []
    cur = dims[ne]
    for j, (k, s, p) in enumerate(dec):
        sk = dims[ne - 1 - j]
        ddims.append((cur, sk))
        cur = ((sk[0] + 2 * p - k) // s + 1, (sk[1] + 2 * p - k) // s + 1)
        sizes += [B * sk[0] * sk[1], B * cur[0] * cur[1], B * cur[0] * cur[1]]
    # one allocation, every map starting on a 256-byte boundary
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot)
        tot += (n + 63) // 64 * 64
    buf = torch.empty(tot, dtype=torch.float32, device=mask.device)
    it = iter(zip(offs, sizes))

    def view(shape):
        o, n = next(it)
        return buf[o:o + n].view(shape)

    m, er = [mask], [None]
    mo_views = [view((B,) + dims[i + 1]) for i in range(ne)]
    r_views = [view((B,) + dims[i + 1]) for i in range(ne)]
    # A level with few pixels per image is pure launch latency: runs of such levels go into ONE launch in which a single
    # workgroup per image walks them (tg_mask_pyramid); a large level (> MASK_FUSE_PIXELS outputs per image) needs the whole
    # chip and keeps its own launch.  At 256x256: enc1 | enc2 ... enc7, dec7 ... dec3 (16 ops) | dec2 x 2 | dec1 x 2.
    pm = L.TgMaskPyramid()
    n_ops = 0
    lib = _lib()

    def flush():
        nonlocal n_ops, pm
        if n_ops:
            pm.nops = n_ops
            L.check(lib.tg_mask_pyramid(C.byref(pm), B, _stream()), "tg_mask_pyramid")
            pm, n_ops = L.TgMaskPyramid(), 0

    def put(kind, hin, hout, k, s, p, a, a2, o, o2):
        nonlocal n_ops
        if hout[0] * hout[1] > MASK_FUSE_PIXELS:
            flush()
            if kind == 0:
                L.check(lib.tg_mask_update(_p(a), B, hin[0], hin[1], k, s, p, hout[0], hout[1], _p(o), _p(o2), _stream()), "tg_mask_update")
            else:
                L.check(lib.tg_mask_up_merge(_p(a), _p(a2), B, hin[0], hin[1], hout[0], hout[1], _p(o), _stream()), "tg_mask_up_merge")
            return
        if n_ops == L.TG_MASK_PYRAMID_MAX:
            flush()
        op = pm.op[n_ops]
        op.kind, op.H, op.W, op.Ho, op.Wo, op.k, op.stride, op.pad = kind, hin[0], hin[1], hout[0], hout[1], k, s, p
        op.in_, op.in2, op.out, op.out2 = a.data_ptr(), (a2.data_ptr() if a2 is not None else None), o.data_ptr(), \
            (o2.data_ptr() if o2 is not None else None)
        n_ops += 1

    for i, (k, s, p) in enumerate(enc):
        put(0, dims[i], dims[i + 1], k, s, p, m[-1], None, mo_views[i], r_views[i])
        m.append(mo_views[i])
        er.append(r_views[i])
    dm, dmasks, dr = m[ne], [], []
    for j, (k, s, p) in enumerate(dec):
        (hin, sk) = ddims[j]
        skip_m = m[ne - 1 - j]
        mm = view((B,) + sk)
        hout = ((sk[0] + 2 * p - k) // s + 1, (sk[1] + 2 * p - k) // s + 1)
        mo, r = view((B,) + hout), view((B,) + hout)
        put(1, hin, sk, 0, 0, 0, dm, skip_m, mm, None)
        put(0, sk, hout, k, s, p, mm, None, mo, r)
        dmasks.append(mm)
        dr.append(r)
        dm = mo
    flush()
    return m, er, dmasks, dr


BN_SMALL_ROWS = 2048          # tg_bn_fwd / tg_bn_act_bwd take their one-launch form up to this many rows (pointwise.hip)


def bn_stats(y, running_mean=None, running_var=None, nbt=None, eps=BN_EPS, momentum=BN_MOMENTUM):
    _chk(y, "y")
    Cc = y.shape[-1]
    rows = y.numel() // Cc
    mean, rstd = empty(Cc, like=y), empty(Cc, like=y)
    lib = _lib()
    ws = workspace(lib.tg_bn_ws_bytes(rows, Cc))
    nbt_p = None if nbt is None else C.c_void_p(nbt.data_ptr())
    L.check(lib.tg_bn_stats(_p(y), rows, Cc, eps, momentum, _p(mean), _p(rstd), _p(running_mean), _p(running_var), nbt_p,
                            _p(ws), ws.numel() * 4, _stream()), "tg_bn_stats")
    return mean, rstd


def bn_fwd(y, gamma, beta, act, slope=0.0, running_mean=None, running_var=None, nbt=None, out=None, eps=BN_EPS,
           momentum=BN_MOMENTUM, apply=True):
    """Training-mode BatchNorm forward: (mean, rstd, out) = bn_stats + bn_act_fwd in one call (one launch on small maps).
    apply=False: statistics (and running-statistics update) only, out = None -- the consumer applies the affine map + activation
    while it loads y (conv_fwd_bnin / conv_wgrad_bnin)."""
    _chk(y, "y"); _chk(out, "out")
    Cc = y.shape[-1]
    rows = y.numel() // Cc
    mean, rstd = empty(Cc, like=y), empty(Cc, like=y)
    if out is None and apply:
        out = torch.empty_like(y)
    lib = _lib()
    ws = workspace(lib.tg_bn_ws_bytes(rows, Cc))
    nbt_p = None if nbt is None else C.c_void_p(nbt.data_ptr())
    L.check(lib.tg_bn_fwd(_p(y), rows, Cc, eps, momentum, _p(gamma.detach()), _p(beta.detach()), act, slope, _p(mean), _p(rstd),
                          _p(running_mean), _p(running_var), nbt_p, _p(out), _p(ws), ws.numel() * 4, _stream()), "tg_bn_fwd")
    return mean, rstd, out


def bn_eval_stats(running_mean, running_var, eps=BN_EPS):
    Cc = running_mean.numel()
    mean, rstd = torch.empty_like(running_mean), torch.empty_like(running_mean)
    L.check(_lib().tg_bn_eval_stats(_p(running_mean), _p(running_var), Cc, eps, _p(mean), _p(rstd), _stream()), "tg_bn_eval_stats")
    return mean, rstd


def bn_running_update(mean, rstd, rows, running_mean, running_var, nbt, eps=BN_EPS, momentum=BN_MOMENTUM):
    nbt_p = None if nbt is None else C.c_void_p(nbt.data_ptr())
    L.check(_lib().tg_bn_running_update(_p(mean), _p(rstd), rows, mean.numel(), eps, momentum, _p(running_mean),
                                        _p(running_var), nbt_p, _stream()), "tg_bn_running_update")


def bn_fwd_grouped(y, groups, gamma, beta, act, slope=0.0, out=None, eps=BN_EPS):
    """Training-mode BatchNorm of `groups` passes stacked along the leading dimension, statistics per pass, in one set of launches
    (tg_bn_fwd_grouped).  -> (mean [groups, C], rstd [groups, C], out).  No running-statistics update (bn_running_update_multi)."""
    _chk(y, "y"); _chk(out, "out")
    Cc = y.shape[-1]
    rows_g = y.numel() // Cc // groups
    assert rows_g * groups * Cc == y.numel() and y.shape[0] % groups == 0
    mean, rstd = empty(groups, Cc, like=y), empty(groups, Cc, like=y)
    if out is None:
        out = torch.empty_like(y)
    lib = _lib()
    ws = workspace(lib.tg_bn_grouped_ws_bytes(rows_g, groups, Cc))
    L.check(lib.tg_bn_fwd_grouped(_p(y), rows_g, groups, Cc, eps, _p(gamma.detach()), _p(beta.detach()), act, slope, _p(mean), _p(rstd),
                                  _p(out), _p(ws), ws.numel() * 4, _stream()), "tg_bn_fwd_grouped")
    return mean, rstd, out


def bn_act_bwd_grouped(dout, y, groups, mean, rstd, gamma, beta, act, slope=0.0, want_dbias=True, outs=None):
    """Backward of bn_fwd_grouped, in place on dout.  -> (dy, dgamma, dbeta, dbias): parameter gradients summed over the passes."""
    _chk(dout, "dout"); _chk(y, "y"); _chk(mean, "mean"); _chk(rstd, "rstd")
    Cc = y.shape[-1]
    rows_g = y.numel() // Cc // groups
    if outs is not None:
        dgamma, dbeta, dbias = outs
    else:
        dgamma, dbeta = empty(Cc, like=y), empty(Cc, like=y)
        dbias = empty(Cc, like=y) if want_dbias else None
    lib = _lib()
    ws = workspace(lib.tg_bn_grouped_ws_bytes(rows_g, groups, Cc))
    L.check(lib.tg_bn_act_bwd_grouped(_p(dout), _p(y), rows_g, groups, Cc, _p(mean), _p(rstd), _p(gamma.detach()), _p(beta.detach()), act,
                                      slope, _p(dout), _p(dgamma), _p(dbeta), _p(dbias), _p(ws), ws.numel() * 4, _stream()),
            "tg_bn_act_bwd_grouped")
    return dout, dgamma, dbeta, dbias


def bn_running_update_multi(mean, rstd, rows_g, order, running_mean, running_var, nbt, eps=BN_EPS, momentum=BN_MOMENTUM):
    """Running-statistics updates of passes `order` (indices into mean / rstd [groups, C]), one after the other, in ONE launch."""
    arr = (C.c_int * len(order))(*order)
    nbt_p = None if nbt is None else C.c_void_p(nbt.data_ptr())
    L.check(_lib().tg_bn_running_update_multi(_p(mean), _p(rstd), rows_g, mean.shape[-1], eps, momentum, arr, len(order),
                                              _p(running_mean), _p(running_var), nbt_p, _stream()), "tg_bn_running_update_multi")


def bn_act_fwd(y, mean, rstd, gamma, beta, act, slope=0.0, out=None):
    _chk(y, "y"); _chk(out, "out")
    Cc = y.shape[-1]
    if out is None:
        out = torch.empty_like(y)
    L.check(_lib().tg_bn_act_fwd(_p(y), y.numel() // Cc, Cc, _p(mean), _p(rstd), _p(gamma.detach()), _p(beta.detach()), act,
                                 slope, _p(out), _stream()), "tg_bn_act_fwd")
    return out


def bn_act_bwd(dout, y, mean, rstd, gamma, beta, act, slope=0.0, ratio=None, inplace=True, want_dbias=True, outs=None):
    """Returns (dy, dgamma, dbeta, dbias); dy overwrites dout when inplace.  dbias = sum_rows dy (the gradient of
    the bias of the conv feeding this BatchNorm), from the same reduction pass."""
    _chk(dout, "dout"); _chk(y, "y"); _chk(ratio, "ratio")
    Cc = y.shape[-1]
    rows = y.numel() // Cc
    dy = dout if inplace else torch.empty_like(dout)
    if outs is not None:                              # (dgamma, dbeta, dbias) persistent gradient buffers
        dgamma, dbeta, dbias = outs
    else:
        dgamma, dbeta = empty(Cc, like=y), empty(Cc, like=y)
        dbias = empty(Cc, like=y) if want_dbias else None
    lib = _lib()
    ws = workspace(lib.tg_bn_ws_bytes(rows, Cc))
    L.check(lib.tg_bn_act_bwd(_p(dout), _p(y), rows, Cc, _p(mean), _p(rstd), _p(gamma.detach()), _p(beta.detach()), act, slope,
                              _p(ratio), _p(dy), _p(dgamma), _p(dbeta), _p(dbias), _p(ws), ws.numel() * 4, _stream()),
            "tg_bn_act_bwd")
    return dy, dgamma, dbeta, dbias


def bn_bwd_conv1_supported(y_shape):
    Cc = y_shape[-1]
    rows = 1
    for d in y_shape[:-1]:
        rows *= d
    return bool(_lib().tg_bn_bwd_conv1_supported(rows, Cc))


def bn_act_bwd_conv1(dz, w, y, mean, rstd, gamma, beta, act, slope=0.0, ratio=None, want_dbias=True, outs=None):
    """bn_act_bwd with dout = conv_dgrad(dz, w) of a C -> 1 channel 3x3 / stride-1 / pad-1 conv recomputed on the fly from the
    1-channel dz ([B][H][W] or [B][H][W][1]); w = that conv's weight.  Returns (dy, dgamma, dbeta, dbias), dy a new tensor."""
    _chk(dz, "dz"); _chk(y, "y"); _chk(ratio, "ratio")
    wv = weight_view(w)
    _chk(wv, "weight")
    B, H, W, Cc = y.shape
    assert dz.numel() == B * H * W and tuple(wv.shape) == (1, 3, 3, Cc), (tuple(dz.shape), tuple(wv.shape))
    dy = torch.empty_like(y)
    if outs is not None:
        dgamma, dbeta, dbias = outs
    else:
        dgamma, dbeta = empty(Cc, like=y), empty(Cc, like=y)
        dbias = empty(Cc, like=y) if want_dbias else None
    lib = _lib()
    ws = workspace(lib.tg_bn_conv1_ws_bytes(B * H * W, Cc))
    L.check(lib.tg_bn_act_bwd_conv1(_p(dz), _p(wv), B, H, W, _p(y), Cc, _p(mean), _p(rstd), _p(gamma.detach()), _p(beta.detach()), act,
                                    slope, _p(ratio), _p(dy), _p(dgamma), _p(dbeta), _p(dbias), _p(ws), ws.numel() * 4, _stream()),
            "tg_bn_act_bwd_conv1")
    return dy, dgamma, dbeta, dbias


def act_bwd(dout, out, act, slope=0.0, ratio=None, inplace=True):
    _chk(dout, "dout"); _chk(out, "out"); _chk(ratio, "ratio")
    Cc = dout.shape[-1]
    din = dout if inplace else torch.empty_like(dout)
    L.check(_lib().tg_act_bwd(_p(dout), _p(out), dout.numel() // Cc, Cc, act, slope, _p(ratio), _p(din), _stream()), "tg_act_bwd")
    return din


def upcat_bn_supported(up_shape, skip_shape, H, W):
    """Can upcat_fwd(..., up_bn=...) apply the BatchNorm + activation of the layer below while it loads `up`?"""
    B, h, w, Cu = up_shape
    Cs = 0 if skip_shape is None else skip_shape[3]
    return bool(_lib().tg_upcat_bn_supported(B, h, w, Cu, H, W, Cs))


def upcat_fwd(up, skip, H, W, out_mask=None, up_bn=None):
    """up_bn = (mean, rstd, gamma, beta[, act[, slope]]): `up` is a PRE-BatchNorm conv output, act(BN(up)) is formed on load."""
    _chk(up, "up"); _chk(skip, "skip"); _chk(out_mask, "out_mask")
    B, h, w, Cu = up.shape
    Cs = 0 if skip is None else skip.shape[3]
    out = empty(B, H, W, Cu + Cs, like=up)
    if up_bn is not None:
        bn = _bn_act(up_bn)
        L.check(_lib().tg_upcat_fwd_bn(_p(up), C.byref(bn), _p(skip), _p(out_mask), B, h, w, Cu, H, W, Cs, _p(out), _stream()),
                "tg_upcat_fwd_bn")
        return out
    L.check(_lib().tg_upcat_fwd(_p(up), _p(skip), _p(out_mask), B, h, w, Cu, H, W, Cs, _p(out), _stream()), "tg_upcat_fwd")
    return out


def upcat_bwd(dout, h, w, Cu, want_skip=True):
    _chk(dout, "dout")
    B, H, W, Ct = dout.shape
    Cs = Ct - Cu
    dup = empty(B, h, w, Cu, like=dout)
    dskip = empty(B, H, W, Cs, like=dout) if (Cs > 0 and want_skip) else None
    L.check(_lib().tg_upcat_bwd(_p(dout), B, h, w, Cu, H, W, Cs, _p(dup), _p(dskip), _stream()), "tg_upcat_bwd")
    return dup, dskip


def sigmoid_composite_fwd(logits, x, mask, out=None):
    _chk(out, "out")
    if out is None:
        out = torch.empty_like(x)
    assert out.shape == x.shape
    L.check(_lib().tg_sigmoid_composite_fwd(_p(logits), _p(x), _p(mask), x.numel(), _p(out), _stream()), "tg_sigmoid_composite_fwd")
    return out


def sigmoid_composite_bwd(dout, logits, mask, want_dx=False):
    _chk(dout, "dout")
    dz = torch.empty_like(logits)
    dx = torch.empty_like(dout) if want_dx else None
    L.check(_lib().tg_sigmoid_composite_bwd(_p(dout), _p(logits), _p(mask), dout.numel(), _p(dz), _p(dx), _stream()),
            "tg_sigmoid_composite_bwd")
    return dz, dx


def maxpool2_fwd(x):
    B, H, W, Cc = x.shape
    out = empty(B, H // 2, W // 2, Cc, like=x)
    L.check(_lib().tg_maxpool2_fwd(_p(x), B, H, W, Cc, _p(out), _stream()), "tg_maxpool2_fwd")
    return out


def maxpool2_bwd(dout, x, relu_gate=False):
    B, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    L.check(_lib().tg_maxpool2_bwd(_p(dout), _p(x), B, H, W, Cc, 1 if relu_gate else 0, _p(dx), _stream()), "tg_maxpool2_bwd")
    return dx


def pixel_losses(pred, target, mask, w_l1, w_tv, w_bnd, l1_weight=None, gscale=None, dpred=None, accumulate=False,
                 want_grad=True, eps=1e-6):
    """-> (out5 device tensor {l1, tv, boundary, sum(band), total}, dpred or None)."""
    _chk(pred, "pred"); _chk(target, "target"); _chk(mask, "mask"); _chk(l1_weight, "l1_weight")
    B, H, W = pred.shape
    out5 = empty(5, like=pred)
    if want_grad and dpred is None:
        dpred = torch.empty_like(pred)
        accumulate = False
    lib = _lib()
    ws = workspace(lib.tg_pixel_loss_ws_bytes(B, H, W))
    L.check(lib.tg_pixel_losses(_p(pred), _p(target), _p(mask), _p(l1_weight), B, H, W, w_l1, w_tv, w_bnd, eps, _p(gscale),
                                _p(out5), _p(dpred) if want_grad else None, 1 if accumulate else 0, _p(ws), ws.numel() * 4,
                                _stream()), "tg_pixel_losses")
    return out5, (dpred if want_grad else None)


def l1_mean(a, b, coef=1.0, gscale=None, want_grad=True, relu_gate=False):
    """relu_gate: `a` is a ReLU output; the gradient returned is the one in front of that ReLU (zero where a <= 0)."""
    _chk(a, "a"); _chk(b, "b")
    out = empty(1, like=a)
    da = torch.empty_like(a) if want_grad else None
    lib = _lib()
    ws = workspace(lib.tg_reduce_ws_bytes(a.numel()))
    fn = lib.tg_l1_mean_relu if relu_gate else lib.tg_l1_mean
    L.check(fn(_p(a), _p(b), a.numel(), coef, _p(gscale), _p(out), _p(da), _p(ws), ws.numel() * 4, _stream()), "tg_l1_mean")
    return out, da


def bce_logits(z, target, coef=1.0, gscale=None, want_grad=True, dz_out=None):
    """dz_out: preallocated destination of the gradient (a slice of a stacked buffer), same shape as z."""
    _chk(z, "z"); _chk(dz_out, "dz_out")
    out = empty(1, like=z)
    if dz_out is not None:
        assert want_grad and dz_out.shape == z.shape
    dz = (dz_out if dz_out is not None else torch.empty_like(z)) if want_grad else None
    lib = _lib()
    ws = workspace(lib.tg_reduce_ws_bytes(z.numel()))
    L.check(lib.tg_bce_logits(_p(z), z.numel(), target, coef, _p(gscale), _p(out), _p(dz), _p(ws), ws.numel() * 4, _stream()),
            "tg_bce_logits")
    return out, dz


QUALITY_KEYS = ("mse", "psnr", "ssim", "l1_distance", "l2_distance", "boundary_mse", "boundary_psnr",
                "boundary_gradient_diff", "boundary_sum")


def quality_metrics(pred, target, mask):
    """-> 9-element device tensor, QUALITY_KEYS order (tg_quality_metrics); pred/target/mask [B,1,H,W] or [B,H,W]."""
    _chk(pred, "pred"); _chk(target, "target"); _chk(mask, "mask")
    H, W = pred.shape[-2], pred.shape[-1]
    imgs = pred.numel() // (H * W)
    assert target.shape == pred.shape and mask.numel() == pred.numel(), (pred.shape, target.shape, mask.shape)
    out = empty(9, like=pred)
    lib = _lib()
    ws = workspace(lib.tg_quality_metrics_ws_bytes(imgs, H, W))
    L.check(lib.tg_quality_metrics(_p(pred), _p(target), _p(mask), imgs, H, W, _p(out), _p(ws), ws.numel() * 4, _stream()),
            "tg_quality_metrics")
    return out


def u8_to_tiles(img_u8=None, mask_u8=None):
    """uint8 device tensors -> (image/255, mask>0) fp32 device tensors of the same shape (tg_u8_to_tiles)."""
    ref = img_u8 if img_u8 is not None else mask_u8
    for t in (img_u8, mask_u8):
        if t is not None and (not t.is_cuda or t.dtype != torch.uint8 or not t.is_contiguous()):
            raise L.TgError("u8_to_tiles: expected contiguous uint8 HIP tensors")
    img = torch.empty(ref.shape, dtype=torch.float32, device=ref.device) if img_u8 is not None else None
    msk = torch.empty(ref.shape, dtype=torch.float32, device=ref.device) if mask_u8 is not None else None
    L.check(_lib().tg_u8_to_tiles(_p(img_u8), _p(mask_u8), ref.numel(), _p(img), _p(msk), _stream()), "tg_u8_to_tiles")
    return img, msk


def _dense_layouts(t):
    """Which dense physical orders a tensor's strides describe: 'c' (row-major) and/or 'cl'."""
    out = set()
    if t.is_contiguous():
        out.add("c")
    if t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous():
        out.add("cl")
    return out


def adam_(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """In-place Adam on the physical storage of p/g/m/v (which must share one dense layout)."""
    lay = _dense_layouts(p)
    for t, nm in ((g, "g"), (m, "m"), (v, "v")):
        if t.shape != p.shape or not (_dense_layouts(t) & lay):
            raise L.TgError(f"adam_: {nm} layout {tuple(t.shape)}/{t.stride()} differs from the parameter's "
                            f"{tuple(p.shape)}/{p.stride()}")
    L.check(_lib().tg_adam(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, _stream()), "tg_adam")
    weights_updated([p])


ADAM_CHUNK = 1 << 14
_adam_tables = {}
adam_scalar_arena = None      # set by tg_hip.graph while a train step is captured / replayed: see AdamScalarArena


class AdamScalarArena:
    """Per-step Adam scalars in device memory (tg_adam_multi_s) for hipGraph replay.  While `active` (during capture),
    every adam_multi_ call takes the next slot and remembers (lr, betas, step).  `refresh(k)` recomputes every slot for `k`
    steps later on the host (tg_adam_scalars: the very two floats the eager launch would have been given) and writes them
    to the device through kernel arguments (tg_write_floats, <= 8 slots per launch) on the current stream."""

    def __init__(self, device, nslots=8):
        import numpy as np
        self.dev = torch.zeros(nslots, 2, dtype=torch.float32, device=device)
        self.host = np.zeros((nslots, 2), dtype=np.float32)
        self.slots, self.active = [], False

    def take(self, lr, beta1, beta2, step):
        i = len(self.slots)
        if i >= self.dev.shape[0]:
            raise L.TgError("AdamScalarArena: more optimiser launches per step than slots")
        self.slots.append((lr, beta1, beta2, int(step)))
        return self.dev[i]

    def refresh(self, k):
        lib = _lib()
        for i, (lr, b1, b2, step) in enumerate(self.slots):
            L.check(lib.tg_adam_scalars(lr, b1, b2, step + k, C.c_void_p(self.host[i].ctypes.data)), "tg_adam_scalars")
        n = 2 * len(self.slots)
        if n:
            L.check(lib.tg_write_floats(_p(self.dev), n, C.c_void_p(self.host.ctypes.data), _stream()), "tg_write_floats")


def adam_multi_(params, grads, ms, vs, lr, beta1, beta2, eps, step, grad_scale=1.0):
    """One-launch Adam over many tensors.  The (p, g, m, v, n) table and the work list live on the device and are
    rebuilt only when a pointer changes (persistent gradient buffers keep them stable step after step)."""
    import numpy as np
    key = tuple(t.data_ptr() for ts in (params, grads, ms, vs) for t in ts)
    ent = _adam_tables.get(key)
    if ent is None:
        lay = None
        for p, g, m, v in zip(params, grads, ms, vs):
            lay = _dense_layouts(p)
            for t, nm in ((g, "g"), (m, "m"), (v, "v")):
                if t.shape != p.shape or not (_dense_layouts(t) & lay):
                    raise L.TgError(f"adam_multi_: {nm} layout differs from the parameter's ({tuple(p.shape)}/{p.stride()})")
        seg = np.zeros((len(params), 5), dtype=np.int64)
        work = []
        for i, (p, g, m, v) in enumerate(zip(params, grads, ms, vs)):
            seg[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
            work += [(i, c) for c in range((p.numel() + ADAM_CHUNK - 1) // ADAM_CHUNK)]
        dev = params[0].device
        ent = (torch.from_numpy(seg).to(dev), torch.tensor(work, dtype=torch.int32).to(dev), len(work))
        if len(_adam_tables) > 64:
            _adam_tables.clear()
        _adam_tables[key] = ent
    segs, work, nwork = ent
    if adam_scalar_arena is not None and adam_scalar_arena.active:
        scal = adam_scalar_arena.take(lr, beta1, beta2, step)
        L.check(_lib().tg_adam_multi_s(C.c_void_p(segs.data_ptr()), C.c_void_p(work.data_ptr()), nwork, ADAM_CHUNK, beta1, beta2,
                                       eps, _p(scal), grad_scale, _stream()), "tg_adam_multi_s")
    else:
        L.check(_lib().tg_adam_multi(C.c_void_p(segs.data_ptr()), C.c_void_p(work.data_ptr()), nwork, ADAM_CHUNK, lr, beta1, beta2,
                                     eps, step, grad_scale, _stream()), "tg_adam_multi")
    weights_updated(params)          # prepared conv weights of these tensors are stale now


def axpby_(x, a, b, y):
    """y = a*x + b*y (same dense layout)."""
    assert x.shape == y.shape and (_dense_layouts(x) & _dense_layouts(y)), (x.shape, x.stride(), y.stride())
    L.check(_lib().tg_axpby(_p(x), a, b, _p(y), y.numel(), _stream()), "tg_axpby")
    return y


def lincomb(x, a, y, b):
    out = torch.empty_like(x)
    L.check(_lib().tg_lincomb(_p(x), a, _p(y), b, _p(out), x.numel(), _stream()), "tg_lincomb")
    return out


def mul(a, b, keep=None):
    """a*b; keep (optional, same shape, contiguous): also receives a copy of `a` from the same pass (tg_mul_keep)."""
    _chk(a, "a"); _chk(b, "b"); _chk(keep, "keep")
    out = torch.empty_like(a)
    if keep is not None:
        assert keep.numel() == a.numel()
        L.check(_lib().tg_mul_keep(_p(a), _p(b), _p(out), _p(keep), a.numel(), _stream()), "tg_mul_keep")
    else:
        L.check(_lib().tg_mul(_p(a), _p(b), _p(out), a.numel(), _stream()), "tg_mul")
    return out


def nchw_to_nhwc(x):
    """Logical [B,C,H,W] contiguous tensor -> [B,H,W,C] contiguous (C==1 is a free reshape)."""
    B, Cc, H, W = x.shape
    x = x if x.is_contiguous() else x.contiguous()
    if Cc == 1:
        return x.reshape(B, H, W, 1)
    y = empty(B, H, W, Cc, like=x)
    L.check(_lib().tg_nchw_to_nhwc(_p(x), B, Cc, H, W, _p(y), _stream()), "tg_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x):
    B, H, W, Cc = x.shape
    if Cc == 1:
        return x.reshape(B, 1, H, W)
    y = empty(B, Cc, H, W, like=x)
    L.check(_lib().tg_nhwc_to_nchw(_p(x), B, Cc, H, W, _p(y), _stream()), "tg_nhwc_to_nchw")
    return y
