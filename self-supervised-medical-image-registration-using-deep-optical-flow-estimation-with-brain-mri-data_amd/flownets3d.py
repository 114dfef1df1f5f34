"""FlowNetS over volumes -- BASELINE config "3D FlowNetS on 128^3 synthetic brain volumes" (SURVEY section 8 row a14).

The reference has no 3-D FlowNetS; this is its 2-D predictor (FlowNetS/FlowNetS.py:10-91, FlowNetS/util.py:17-55) with every
Conv2d / BatchNorm2d / ConvTranspose2d replaced by its 3-D counterpart, three flow channels (x, y, z displacement) and the
same module names, wrapped like reference models.py:209-292 (`opticalFlowReg`): predictor -> per-scale warp of the moving
volume.  All contractions run on the depth-enabled LDS-DMA implicit-GEMM kernels (forward, backward-data per parity class,
backward-weights per depth tap), BatchNorm3d + LeakyReLU on the row kernels of the 2-D path, the tail on csrc/volume_ops.hip.
Training goes through torch.autograd: the predictor is one autograd function whose backward is the HIP backward pass.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import _lib
from .affine3d import Conv3dLayer, Vol
from .engine import BatchNormAct, WoptJob, Workspace, _stream, assign_tiles, rup, upload_table, zero_tensors
from .volume import resize_trilinear, stn3d

WGRAD_SIDE_STREAM = os.environ.get("MIREG_3D_WGRAD_MAIN", "0") != "1"
SIDE_IN_GRAPH = os.environ.get("MIREG_3D_SIDE_IN_GRAPH", "0") == "1"          # experiment: backward-weights branch also inside a capture
TINY_UPSAMPLERS = os.environ.get("MIREG_3D_GEMM_UPSAMPLERS", "0") != "1"      # A/B switch: the flow upsamplers on the GEMM path

ENC = [("conv1", 7, 2), ("conv2", 5, 2), ("conv3", 5, 2), ("conv3_1", 3, 1), ("conv4", 3, 2), ("conv4_1", 3, 1),
       ("conv5", 3, 2), ("conv5_1", 3, 1), ("conv6", 3, 2), ("conv6_1", 3, 1)]


def _conv(cin: int, cout: int, k: int, s: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv3d(cin, cout, k, s, (k - 1) // 2, bias=False), nn.BatchNorm3d(cout), nn.LeakyReLU(0.1, inplace=True))


def _deconv(cin: int, cout: int) -> nn.Sequential:
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, 4, 2, 1, bias=False), nn.LeakyReLU(0.1, inplace=True))


class _PredictorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, *params):
        ctx.mod = mod
        return tuple(mod._forward_impl(x, keep=True))

    @staticmethod
    def backward(ctx, *gflows):
        return (None, None, *ctx.mod._backward_impl(gflows))


class FlowNetS3D(nn.Module):
    """FlowNetS (batch-norm variant, the one the reference trains: models.py:252) over volumes.  `width_div` scales every
    channel width down (tests); volumes must be divisible by 64 per axis (six stride-2 stages, no crop_like needed)."""

    def __init__(self, precision: str = "bf16", width_div: int = 1):
        super().__init__()
        self.precision = precision
        c = [max(8, v // width_div) for v in (64, 128, 256, 512, 512, 1024)]
        d = [max(8, v // width_div) for v in (512, 256, 128, 64)]                   # deconv5..deconv2 outputs
        self.c, self.d = c, d
        cin = [2, c[0], c[1], c[2], c[2], c[3], c[3], c[4], c[4], c[5]]
        cout = [c[0], c[1], c[2], c[2], c[3], c[3], c[4], c[4], c[5], c[5]]
        for (name, k, s), ci, co in zip(ENC, cin, cout):
            setattr(self, name, _conv(ci, co, k, s))
        self.cat = {5: c[4] + d[0] + 3, 4: c[3] + d[1] + 3, 3: c[2] + d[2] + 3, 2: c[1] + d[3] + 3}
        self.deconv5 = _deconv(c[5], d[0])
        self.deconv4 = _deconv(self.cat[5], d[1])
        self.deconv3 = _deconv(self.cat[4], d[2])
        self.deconv2 = _deconv(self.cat[3], d[3])
        self.predict_flow6 = nn.Conv3d(c[5], 3, 3, 1, 1, bias=False)
        for lv in (5, 4, 3, 2):
            setattr(self, f"predict_flow{lv}", nn.Conv3d(self.cat[lv], 3, 3, 1, 1, bias=False))
        for a, b in ((6, 5), (5, 4), (4, 3), (3, 2)):
            setattr(self, f"upsampled_flow{a}_to_{b}", nn.ConvTranspose3d(3, 3, 4, 2, 1, bias=False))
        for m in self.modules():                                                     # FlowNetS/FlowNetS.py:45-52
            if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                nn.init.kaiming_normal_(m.weight, 0.1)
            elif isinstance(m, nn.BatchNorm3d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._eng: Dict[tuple, dict] = {}
        self._last = None
        self._fopt, self._fidx, self._fe = None, {}, None             # packed-domain optimizer (mireg.Adam(fuse=...))

    # ---- engine -----------------------------------------------------------------------------------------------------
    def _engine(self, x: torch.Tensor) -> dict:
        dtype = torch.bfloat16 if self.precision == "bf16" else torch.float32
        key = (tuple(x.shape), x.device, dtype, self.conv1[0].weight.data_ptr())
        if key in self._eng:
            return self._eng[key]
        self._eng.clear()
        B, C, D, H, W = x.shape
        if C != 2 or D % 64 or H % 64 or W % 64:
            raise RuntimeError(f"FlowNetS3D expects (B, 2, D, H, W) volumes divisible by 64 per axis, got {tuple(x.shape)}")
        dev, ws = x.device, Workspace(x.device, dtype)
        dims = {0: (D, H, W)}
        for lv in range(1, 7):
            dims[lv] = tuple(v // 2 for v in dims[lv - 1])
        c, d = self.c, self.d

        def buf(lv: int, ch: int, dt=dtype) -> Vol:
            return Vol(torch.zeros(B, *dims[lv], rup(ch, 8), device=dev, dtype=dt), dims[lv], ch)

        # conv1 (two input channels, 7^3 taps) runs as a (7, 7, 1) convolution over the x-axis im2col of the input:
        # (tx, ci) -> 14 of 16 channels instead of 2 of 8, so 4x less gathered bytes and MFMA work (mireg_stem3d_gather)
        sdims = (D, H, W // 2)
        x0 = Vol(torch.zeros(B, *sdims, 16, device=dev, dtype=dtype), sdims, 16)
        c1 = self.conv1[0].out_channels
        e = dict(ws=ws, dims=dims, x0=x0, w1s=torch.zeros(c1, 16, 7, 7, 1, device=dev, dtype=torch.float32))
        cat = {lv: buf(lv, self.cat[lv]) for lv in (5, 4, 3, 2)}
        gcat = {lv: buf(lv, self.cat[lv]) for lv in (5, 4, 3, 2)}
        # activation (post BatchNorm + LeakyReLU) of every encoder layer, its level, and where its gradient lives
        lvl = dict(conv1=1, conv2=2, conv3=3, conv3_1=3, conv4=4, conv4_1=4, conv5=5, conv5_1=5, conv6=6, conv6_1=6)
        skip = dict(conv2=2, conv3_1=3, conv4_1=4, conv5_1=5)                        # layers whose output is a concat slice
        act, gact, raw, graw, layers, bns = {}, {}, {}, {}, {}, {}
        for name, k, s in ENC:
            conv, bn = getattr(self, name)[0], getattr(self, name)[1]
            co = conv.out_channels
            if name in skip:
                act[name], gact[name] = cat[skip[name]].slice(0, co), gcat[skip[name]].slice(0, co)
            else:
                act[name], gact[name] = buf(lvl[name], co), buf(lvl[name], co)
            raw[name], graw[name] = buf(lvl[name], co), buf(lvl[name], co)
            if name == "conv1":
                layers[name] = Conv3dLayer(e["w1s"], None, (2, 2, 1), (3, 3, 0), ws)
            else:
                layers[name] = Conv3dLayer(conv.weight, None, (s, s, s), ((k - 1) // 2,) * 3, ws)
            bns[name] = BatchNormAct(bn, ws, 0.1)
        for lv in (6, 5, 4, 3, 2):
            layers[f"predict_flow{lv}"] = Conv3dLayer(getattr(self, f"predict_flow{lv}").weight, None, (1, 1, 1), (1, 1, 1), ws)
        for lv in (5, 4, 3, 2):
            # ConvTranspose3d weight [Cin][Cout][4^3] == Conv3d weight [Co][Ci][4^3] of the adjoint (stride 2, pad 1) convolution
            layers[f"deconv{lv}"] = Conv3dLayer(getattr(self, f"deconv{lv}")[0].weight, None, (2, 2, 2), (1, 1, 1), ws)
            layers[f"up{lv}"] = Conv3dLayer(getattr(self, f"upsampled_flow{lv + 1}_to_{lv}").weight, None, (2, 2, 2), (1, 1, 1), ws)
        e.update(cat=cat, gcat=gcat, act=act, gact=gact, raw=raw, graw=graw, layers=layers, bns=bns,
                 flow={lv: buf(lv, 3) for lv in (6, 5, 4, 3, 2)}, gflow={lv: buf(lv, 3) for lv in (6, 5, 4, 3, 2)},
                 flow32={lv: buf(lv, 3, torch.float32) for lv in (6, 5, 4, 3, 2)})
        self._eng[key] = e
        return e

    def _named(self) -> List[Tuple[str, nn.Parameter]]:
        return list(self.named_parameters())

    @staticmethod
    def _up_forward(up, coarse: Vol, cdims, fine: Vol, fdims, ws: Workspace) -> None:
        """The 3 -> 3 channel ConvTranspose3d flow upsampler: one voxel-parallel launch (`mireg_tiny_deconv3d_fwd`) instead of eight
        parity-class GEMM launches that fill 3 of 128 tile columns."""
        if TINY_UPSAMPLERS:
            _lib.call("mireg_tiny_deconv3d_fwd", coarse.ptr, coarse.ld, up.weight.data_ptr(), fine.ptr, fine.ld, coarse.B, *cdims, ws.code, _stream())
        else:
            up.dgrad(coarse, cdims, fine, fdims)

    @staticmethod
    def _conv_of(L: dict) -> dict:
        """parameter name -> engine layer of every convolution weight."""
        conv_of = {f"{name}.0.weight": L[name] for name, _, _ in ENC}
        for lv in (6, 5, 4, 3, 2):
            conv_of[f"predict_flow{lv}.weight"] = L[f"predict_flow{lv}"]
        for lv in (5, 4, 3, 2):
            conv_of[f"deconv{lv}.0.weight"] = L[f"deconv{lv}"]
            conv_of[f"upsampled_flow{lv + 1}_to_{lv}.weight"] = L[f"up{lv}"]
        return conv_of

    # ---- packed-domain optimizer hook (mireg.Adam(fuse=module)) ---------------------------------------------------------
    def fuse_optimizer(self, opt, index: Dict[int, int]) -> None:
        """Every convolution weight but conv1's (whose engine layout is the x-axis im2col of the stem) is updated by `mireg_adam_pack`
        straight from its backward-weights slab; `index` maps id(parameter) to the optimizer's parameter number."""
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise RuntimeError("mireg.Adam(fuse=...) updates the convolution weights from this rank's gradient slabs and leaves no `.grad` "
                               "for an all-reduce: use it in single-process training only (DistributedDataParallel needs the plain optimizer)")
        self._fopt = opt
        self._fidx = {n: index[id(p)] for n, p in self._named()
                      if p.dim() == 5 and n != "conv1.0.weight" and id(p) in index}

    def fused_pending(self) -> bool:
        return self._fe is not None and bool(self._fe.get("slab_pending"))

    def fused_step(self, opt, tick: int) -> None:
        e = self._fe
        L, ws, st = e["layers"], e["ws"], _stream()
        conv_of, params = self._conv_of(L), dict(self._named())
        red, jobs, r, units, max_taps = [], [], 0, 0, 1
        for pname, i in self._fidx.items():
            lay = conv_of[pname]
            taps = lay.kd * lay.kh * lay.kw
            j = WoptJob()
            j.slab = j.g = lay.slab.data_ptr()
            j.slab_stride, j.nsplit = lay.Co * lay.Kf, lay.slab.shape[0]
            j.Co, j.Ci, j.taps, j.Cpad, j.ld = lay.Co, lay.Ci, taps, lay.Cip, lay.Kf
            j.p, j.F = params[pname].data_ptr(), lay.packF.data_ptr()
            j.m, j.v = opt.state_ptrs(i)
            j.unit0, j.runit0 = units, r
            units += lay.Co * ((lay.Cip + 63) // 64)
            max_taps = max(max_taps, taps)
            jobs.append(j)
            if lay.slab.shape[0] > 1:                                                # split-K slabs: summed in place first (fixed order)
                k = WoptJob.from_buffer_copy(j)
                red.append(k)
                r += (lay.Co * lay.Kf + 255) // 256
        if red:
            e["_ftab_r"] = upload_table(red, ws.device)
            _lib.call("mireg_wgrad_reduce", e["_ftab_r"].data_ptr(), len(red), r, st)
        e["_ftab"] = upload_table(jobs, ws.device)
        _lib.call("mireg_adam_pack", e["_ftab"].data_ptr(), len(jobs), units, max_taps, opt.step_dev.data_ptr(), int(tick), opt.lr,
                  opt.betas[0], opt.betas[1], opt.eps, 1.0, ws.code, st)
        e["slab_pending"] = False
        # the forward packs now hold the new weights: valid for the next forward unless someone writes the parameters in between
        e["fresh"] = [(params[n], params[n]._version) for n in self._fidx]

    def forward(self, x: torch.Tensor):
        """Training: (flow0, flow2, flow3, flow4, flow5, flow6); eval: (flow0, flow2) -- FlowNetS/FlowNetS.py:83-88 with the
        full-resolution flow0 = trilinear upsample of flow2 (align_corners=False)."""
        if not x.is_cuda:
            raise RuntimeError("mireg.FlowNetS3D runs on the MI355X only; there is no CPU fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            flows = _PredictorFn.apply(self, x, *[p for _, p in self._named()])
        else:
            flows = tuple(self._forward_impl(x, keep=False))
        flow0 = resize_trilinear(flows[0], tuple(x.shape[2:]), False)
        return (flow0, *flows) if self.training else (flow0, flows[0])

    def _forward_impl(self, x: torch.Tensor, keep: bool):
        e = self._engine(x)
        ws, st, dims, L = e["ws"], _stream(), e["dims"], e["layers"]
        B, _, D, H, W = x.shape
        x = x.detach().float().contiguous()
        w1 = self.conv1[0].weight.detach()                                          # [co][ci][tz][ty][tx] -> [co][tx*2 + ci][tz][ty][1]
        e["w1s"][:, :14] = w1.permute(0, 4, 1, 2, 3).reshape(w1.shape[0], 14, 7, 7, 1)
        fresh = e.pop("fresh", None)                                                 # set by fused_step: forward packs already rewritten
        fresh = fresh is not None and all(p._version == v for p, v in fresh)
        fused_layers = {id(l) for n, l in self._conv_of(L).items() if n in self._fidx} if fresh else set()
        jobs = [l.pack_job() for l in L.values() if id(l) not in fused_layers]
        units, dunits = assign_tiles(jobs, False)
        tab = upload_table(jobs, x.device)
        _lib.call("mireg_pack_weights", tab.data_ptr(), len(jobs), units, dunits, ws.code, st)
        # backward-data packs: the deconvolutions / flow upsamplers run that form forward, every other layer (bar conv1) backward
        need = [l for n, l in L.items() if n != "conv1" and (keep or n.startswith(("deconv", "up")))]
        e["_tab3"] = Conv3dLayer.pack_dgrad_table(need, ws, from_fwd=True)              # every forward pack is current at this point
        _lib.call("mireg_stem3d_gather", x.data_ptr(), e["x0"].ptr, B, 2, D, H, W, 7, 2, 3, 16, ws.code, st)
        src, training = e["x0"], self.training
        for name, k, s in ENC:
            lay, out = L[name], e["act"][name]
            lay.run(src, src.dims, e["raw"][name], 1.0)
            e["bns"][name].forward(e["raw"][name].view2d(), out.view2d(), training)
            src = out
        c, d, cat = self.c, self.d, e["cat"]
        feat = e["act"]["conv6_1"]
        for lv in (6, 5, 4, 3, 2):
            L[f"predict_flow{lv}"].run(feat, dims[lv], e["flow"][lv], 1.0, y32=e["flow32"][lv])
            if lv == 2:
                break
            nxt, dm = cat[lv - 1], d[6 - lv]                                         # deconv{lv-1} has d[5 - (lv-1)] outputs
            skipc = nxt.C - 3 - dm                                                   # channels of the encoder skip in the next concat
            # deconv (LeakyReLU 0.1) and the flow upsampler are backward-data-form launches into their concat slices
            L[f"deconv{lv - 1}"].dgrad(feat, dims[lv], nxt.slice(skipc, dm), dims[lv - 1], slope=0.1)
            self._up_forward(L[f"up{lv - 1}"], e["flow"][lv], dims[lv], nxt.slice(skipc + dm, 3), dims[lv - 1], ws)
            feat = nxt
        self._last = dict(e=e, B=B) if keep else None
        # logical (B, 3, d, h, w) views of the channel-last fp32 flows (valid until the next forward of this module)
        return [e["flow32"][lv].buf[..., :3].permute(0, 4, 1, 2, 3) for lv in (2, 3, 4, 5, 6)]

    def _backward_impl(self, gflows):
        if self._last is None:
            raise RuntimeError("FlowNetS3D backward without a saved forward (one forward in flight per module)")
        e, B = self._last["e"], self._last["B"]
        self._last = None
        ws, st, dims, L, d = e["ws"], _stream(), e["dims"], e["layers"], self.d
        cat, gcat, act, gact = e["cat"], e["gcat"], e["act"], e["gact"]
        dev = e["x0"].buf.device
        for g, lv in zip(gflows, (2, 3, 4, 5, 6)):
            gf = e["gflow"][lv]
            if g is None:
                zero_tensors([gf.buf])
                continue
            g = g.float().contiguous()
            nv = dims[lv][0] * dims[lv][1] * dims[lv][2]
            _lib.call("mireg_nchw_to_nhwc", g.data_ptr(), gf.ptr, B, 3, 0, 3, nv, gf.ld, ws.code, st)

        def mask(g: Vol, a: Vol, slope: float) -> None:
            _lib.call("mireg_lrelu_bwd", g.ptr, g.ld, a.ptr, a.ld, g.rows, g.C, slope, ws.code, st)

        # backward-weights launches go to a second stream (their inputs are final once the main chain reaches them, nothing on the main
        # chain reads their slabs before the join below): they fill the CUs the backward-data chain leaves idle on the coarse levels
        main = torch.cuda.current_stream()
        use_side = WGRAD_SIDE_STREAM and (SIDE_IN_GRAPH or not torch.cuda.is_current_stream_capturing())   # measured: as a graph branch it costs 0.3 ms
        side = e.get("side") if use_side else None
        if use_side and side is None:
            side = e["side"] = torch.cuda.Stream(device=dev)

        def wgrad(lay, *args) -> None:
            if side is None:
                lay.wgrad(*args)
                return
            side.wait_stream(main)
            with torch.cuda.stream(side):
                lay.wgrad(*args)

        # ---- decoder, fine to coarse ----
        for lv in (2, 3, 4, 5, 6):
            feat = cat[lv] if lv < 6 else act["conv6_1"]
            gfeat = gcat[lv] if lv < 6 else gact["conv6_1"]
            pf = L[f"predict_flow{lv}"]
            wgrad(pf, feat, dims[lv], e["gflow"][lv], dims[lv])
            pf.dgrad(e["gflow"][lv], dims[lv], gfeat, dims[lv], accumulate=(lv > 2))   # level 2 opens gcat2, deeper levels add
            if lv == 6:
                break
            dm = d[5 - lv]                                                           # outputs of deconv{lv}
            skipc = feat.C - 3 - dm
            gup, gdec = gfeat.slice(skipc + dm, 3), gfeat.slice(skipc, dm)
            up, dec = L[f"up{lv}"], L[f"deconv{lv}"]
            below = cat[lv + 1] if lv + 1 < 6 else act["conv6_1"]
            gbelow = gcat[lv + 1] if lv + 1 < 6 else gact["conv6_1"]
            # flow upsampler (ConvTranspose3d 3 -> 3): d/d flow_{lv+1} adds to its loss gradient; weight gradient
            if TINY_UPSAMPLERS:                                                    # voxel-parallel kernels (csrc/volume_ops.hip), main stream
                gc_ = e["gflow"][lv + 1]
                _lib.call("mireg_tiny_deconv3d_bwd_data", gup.ptr, gup.ld, up.weight.data_ptr(), gc_.ptr, gc_.ld, 1, B, *dims[lv + 1], ws.code, st)
                nb = _lib.lib().mireg_tiny_deconv3d_blocks(B, *dims[lv + 1])
                if getattr(up, "slab", None) is None or up.slab.shape[0] != nb:
                    up.slab = torch.zeros(nb, up.Co, up.Kf, device=dev, dtype=torch.float32)
                fc = e["flow"][lv + 1]
                _lib.call("mireg_tiny_deconv3d_bwd_weights", gup.ptr, gup.ld, fc.ptr, fc.ld, up.slab.data_ptr(), nb, up.Cip, B, *dims[lv + 1],
                          ws.code, st)
            else:
                up.run(gup, dims[lv], e["gflow"][lv + 1], 1.0, accumulate=True)
                wgrad(up, gup, dims[lv], e["flow"][lv + 1], dims[lv + 1])
            # deconv: LeakyReLU mask, then backward-data (conv form of the adjoint) opens the gradient of the level below
            mask(gdec, feat.slice(skipc, dm), 0.1)
            dec.run(gdec, dims[lv], gbelow, 1.0)
            wgrad(dec, gdec, dims[lv], below, dims[lv + 1])
        # ---- encoder, deep to shallow ----
        prev = {n: (ENC[i - 1][0] if i else None) for i, (n, _, _) in enumerate(ENC)}
        # BatchNorm scale / shift gradients: where the parameter already has a `.grad` (mireg.Adam.zero_grad keeps and zeroes them) the kernel
        # adds into it directly and autograd gets None -- otherwise every step pays a copy into a hand-over buffer plus AccumulateGrad's add
        # for each of the 20 small tensors (0.2 ms of launch-sized kernels on the serial chain)
        direct = set()
        # (single process only: DistributedDataParallel reduces a gradient when autograd hands it over, which this shortcut skips)
        solo = not (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1)
        own = e.setdefault("bn_own", {n: (b.grad_g, b.grad_b) for n, b in e["bns"].items()})
        for name, _, _ in ENC:
            mod, b = getattr(self, name)[1], e["bns"][name]
            gw, gb = mod.weight.grad, mod.bias.grad
            ok = solo and all(g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.device == dev for g in (gw, gb))
            b.grad_g, b.grad_b = (gw, gb) if ok else own[name]
            if ok:
                direct.add(name)
        for name, k, s in reversed(ENC):
            lay, bn = L[name], e["bns"][name]
            bn.backward(e["raw"][name].view2d(), gact[name].view2d(), e["graw"][name].view2d(), acc_param_grads=name in direct)
            src = act[prev[name]] if prev[name] else e["x0"]
            odims = e["raw"][name].dims
            wgrad(lay, src, src.dims, e["graw"][name], odims)
            if prev[name]:
                # skip tensors already hold the decoder's contribution (they are concat slices): accumulate there
                lay.dgrad(e["graw"][name], odims, gact[prev[name]], src.dims, accumulate=prev[name] in ("conv2", "conv3_1", "conv4_1", "conv5_1"))
        if side is not None:
            main.wait_stream(side)
        # ---- gradients in named_parameters() order ----
        grads, pairs = [], []
        gbuf = e.setdefault("gbuf", {})                       # persistent gradient buffers: stable pointers = cached job tables, capturable

        def gb(key, like, param=None):
            """Two buffers per gradient: autograd ADDS a returned gradient onto an existing .grad, so the buffer handed back must never
            be the tensor .grad currently aliases (it adopts the returned tensor when .grad was None)."""
            pair = gbuf.setdefault(key, [None, None])
            i = 1 if (param is not None and param.grad is not None and pair[0] is not None
                      and param.grad.data_ptr() == pair[0].data_ptr()) else 0
            if pair[i] is None:
                pair[i] = torch.empty_like(like, dtype=torch.float32)
            return pair[i]
        conv_of = self._conv_of(L)
        if self._fidx:
            if e.get("slab_pending"):
                raise RuntimeError("FlowNetS3D: a second backward before optimizer.step(): with mireg.Adam(fuse=...) the weight gradients "
                                   "live in the backward-weights slabs, which one backward fills and one step consumes")
            e["slab_pending"], self._fe = True, e
        for pname, p in self._named():
            if pname in self._fidx:                                                  # updated from its slab by fused_step
                grads.append(None)
                continue
            if pname == "conv1.0.weight":
                g = e["g1s"] = gb("g1s", e["w1s"])                                    # gradient in the stem layout, mapped back below
                pairs.append((conv_of[pname], g))
            elif pname in conv_of:
                g = gb(pname, p, p)                                    # every element is written by the unpack
                pairs.append((conv_of[pname], g))
            else:                                                                    # BatchNorm3d weight / bias
                lname, _, kind = pname.split(".")
                if lname in direct:                                                  # already added into p.grad by the kernel
                    grads.append(None)
                    continue
                g = gb(pname, p, p)
                g.copy_(e["bns"][lname].grad_g if kind == "weight" else e["bns"][lname].grad_b)
            grads.append(g)
        e["_tab"] = Conv3dLayer.unpack_grads(pairs, ws)
        names = [n for n, _ in self._named()]
        c1 = e["g1s"].shape[0]                                                        # [co][tx*2 + ci][tz][ty][1] -> [co][ci][tz][ty][tx]
        g1 = gb("conv1.0.weight", self.conv1[0].weight, self.conv1[0].weight)
        g1.copy_(e["g1s"][:, :14, :, :, 0].reshape(c1, 7, 2, 7, 7).permute(0, 2, 3, 4, 1))
        grads[names.index("conv1.0.weight")] = g1
        return grads


class opticalFlowReg3d(nn.Module):
    """reference models.opticalFlowReg (models.py:209-292) over volumes: `(x) -> (flows, warped)` with x = (B, 2, D, H, W) =
    [fixed, moving]; every flow scale warps the moving volume (`stn3d`); `forward(x, segs)` also returns the warped, rounded
    moving segmentation for `mireg.dice_batch` / `mireg.dice_average` (dimension-agnostic), as the 2-D wrapper does."""

    def __init__(self, precision: str = "bf16", width_div: int = 1):
        super().__init__()
        self.predictor = FlowNetS3D(precision, width_div)

    stn = staticmethod(stn3d)

    def forward(self, x: torch.Tensor, segs: torch.Tensor = None):
        """-> (flows, warped) or, with segs = (B, 2, D, H, W) label volumes [fixed, moving], (flows, warped, warped_segs_int):
        the moving labels ride the finest flow and are rounded / clipped to {0..3} on device (reference models.py:275-278,286)."""
        flows = self.predictor(x)
        moving = x[:, 1:2]
        warped = [stn3d(f, moving) for f in flows]
        if segs is None:
            return list(flows), warped
        from .ops import seg_round
        wseg = seg_round(stn3d(flows[0].detach(), segs[:, 1:2].float()))
        return list(flows), warped, wseg
