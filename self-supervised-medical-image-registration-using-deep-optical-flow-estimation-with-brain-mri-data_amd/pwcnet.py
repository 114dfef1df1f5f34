"""PWC-DC-Net (reference PWC/models/PWCNet.py:38-279, 1-channel pyramid) on the HIP engine.

Same submodule names / state_dict keys as the reference; returns the 7 flows (256^2 .. 4^2 for a 256^2
input) in both train and eval mode.  Every torch.cat of the DenseNet-style estimators is
replaced by channel-offset writes: each level owns ONE buffer laid out
    [conv_4 32 | conv_3 64 | conv_2 96 | conv_1 128 | conv_0 128 | corr 81 | c1 C_l | up_flow 2 | up_feat 2]
so that conv_j reads the suffix that starts at its own output's end (all suffix offsets are multiples of 32).
The backward zeroes every gradient buffer once and lets all producers accumulate (dense connections fan gradients
into overlapping channel ranges); cost-volume and warp backward run on csrc/correlation.hip.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .correlation import Correlation, WarpBwdWorkspace, correlation_bwd_views, correlation_views, pwc_warp_bwd_views, pwc_warp_views
from .engine import F32, View, _stream, cast_from_f32, lrelu_bwd, nchw_to_view, zero_many_table, zero_tensors
from .flownets import drop_engines, grads_for_autograd, PackedOptimizerHook, PredictorEngineBase

SLOPE = 0.1
PYRAMID = [(1, 16), (16, 32), (32, 64), (64, 96), (96, 128), (128, 196)]      # PWCNet.py:50-67
PYR_NAMES = {1: ("conv1a", "conv1aa", "conv1b"), 2: ("conv2a", "conv2aa", "conv2b"), 3: ("conv3a", "conv3aa", "conv3b"),
             4: ("conv4a", "conv4aa", "conv4b"), 5: ("conv5a", "conv5aa", "conv5b"), 6: ("conv6aa", "conv6a", "conv6b")}
DENSE = [128, 128, 96, 64, 32]                                                  # PWCNet.py:73-80
DENSE_OFF = [320, 192, 96, 32, 0]                                               # where conv_j writes; it reads [off+cout:]
BASE = 448                                                                       # start of [corr | c1 | up_flow | up_feat]
FLOW_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}                                 # PWCNet.py:214-258
FEAT_C = {6: 196, 5: 128, 4: 96, 3: 64, 2: 32}
DC = [(128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1)]                  # PWCNet.py:128-133


def _conv(cin, cout, k=3, stride=1, padding=1, dilation=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, padding, dilation, bias=True), nn.LeakyReLU(SLOPE))


class PWCEngine(PredictorEngineBase):
    def __init__(self, module: "PWCDCNet", B: int, H: int, W: int, device, dtype: torch.dtype):
        super().__init__(module, B, H, W, device, dtype)
        if H % 64 or W % 64:
            raise RuntimeError(f"PWC engine needs H, W divisible by 64, got {H}x{W}")
        m, ws, new = module, self.ws, self.ws.new
        self.md = m.md
        self.nd = (2 * m.md + 1) ** 2
        hs = {lvl: (H >> lvl, W >> lvl) for lvl in range(0, 7)}
        self.hs = hs
        for lvl in range(1, 7):
            for i, n in enumerate(PYR_NAMES[lvl]):
                self.add_conv(n, getattr(m, n)[0], 2 if i == 0 else 1, 1, uses=2)     # siamese: one wgrad slot per stream
        for lvl in (6, 5, 4, 3, 2):
            for j in range(5):
                self.add_conv(f"conv{lvl}_{j}", getattr(m, f"conv{lvl}_{j}")[0], 1, 1)
            self.add_conv(f"predict_flow{lvl}", getattr(m, f"predict_flow{lvl}"), 1, 1)
            if lvl > 2:
                self.add_conv(f"deconv{lvl}", getattr(m, f"deconv{lvl}"), 2, 1)
                self.add_conv(f"upfeat{lvl}", getattr(m, f"upfeat{lvl}"), 2, 1)
        self.add_conv("deconv2", m.deconv2, 2, 1)
        self.add_conv("deconv1", m.deconv1, 2, 1)
        self._unused_tail = list(m.deconv0.parameters())     # defined, unused upstream (PWCNet.py:126,274): inside the tail bucket
        for i, (_, d) in enumerate(DC, start=1):
            self.add_conv(f"dc_conv{i}", getattr(m, f"dc_conv{i}")[0], 1, d, d)
        self.add_conv("dc_conv7", m.dc_conv7, 1, 1)
        # ---- buffers ---------------------------------------------------------------------------------
        self.img = {s: new(B, H, W, 1) for s in "ab"}
        self.pyr = {(lvl, s, i): new(B, *hs[lvl], PYRAMID[lvl - 1][1]) for lvl in range(1, 7) for s in "ab" for i in range(3)}
        self.x = {lvl: new(B, *hs[lvl], BASE + self.nd + (0 if lvl == 6 else FEAT_C[lvl] + 4)) for lvl in (6, 5, 4, 3, 2)}
        self.warped = {lvl: new(B, *hs[lvl], FEAT_C[lvl]) for lvl in (5, 4, 3, 2)}
        self.flow32 = {lvl: new(B, *hs[lvl], 2, dtype=F32, pad=2) for lvl in range(0, 7)}
        self.flowT = {lvl: new(B, *hs[lvl], 2) for lvl in range(1, 7)}
        self.upflow32 = {lvl: new(B, *hs[lvl], 2, dtype=F32, pad=2) for lvl in (5, 4, 3, 2)}
        self.dc = [new(B, *hs[2], c) for c, _ in DC]
        self.dc7 = new(B, *hs[2], 2)
        self.grads_ready = False

    def forward(self, x: torch.Tensor, training: bool) -> List[torch.Tensor]:
        L, code, st = self.layers, self.ws.code, _stream()
        self.training_cache = training
        self.pack_weights()
        x = x.contiguous()
        nchw_to_view(x, 0, 1, self.img["a"])
        nchw_to_view(x, 1, 1, self.img["b"])
        for s in "ab":                                         # siamese feature pyramid (PWCNet.py:186-197)
            src = self.img[s]
            for lvl in range(1, 7):
                for i, n in enumerate(PYR_NAMES[lvl]):
                    dst = self.pyr[(lvl, s, i)]
                    L[n].run_fwd_form(src, dst, slope=SLOPE)
                    src = dst
        feat = lambda lvl, s: self.pyr[(lvl, s, 2)]
        for lvl in (6, 5, 4, 3, 2):
            X = self.x[lvl]
            c1, c2 = feat(lvl, "a"), feat(lvl, "b")
            od = self.nd + (0 if lvl == 6 else FEAT_C[lvl] + 4)
            if lvl == 6:
                correlation_views(c1, c2, X.slice(BASE, self.nd), FEAT_C[6], self.md, 1, SLOPE, code)
            else:
                # up_flow / up_feat were written into X by the previous level; warp c2 with the scaled up_flow
                pwc_warp_views(c2, self.upflow32[lvl], FLOW_SCALE[lvl], self.warped[lvl], code)
                correlation_views(c1, self.warped[lvl], X.slice(BASE, self.nd), FEAT_C[lvl], self.md, 1, SLOPE, code)
                _lib.call("mireg_copy_channels", c1.ptr, c1.ld, X.slice(BASE + self.nd, FEAT_C[lvl]).ptr, X.ld, c1.rows,
                          FEAT_C[lvl], 0, code, st)
            for j in range(5):                                 # DenseNet estimator: conv_j reads the suffix, writes in front
                off, cout = DENSE_OFF[j], DENSE[j]
                L[f"conv{lvl}_{j}"].run_fwd_form(X.slice(off + cout, X.C - off - cout), X.slice(off, cout), slope=SLOPE)
            L[f"predict_flow{lvl}"].run_fwd_form(X, self.flowT[lvl], y32=self.flow32[lvl])
            if lvl > 2:
                nxt = self.x[lvl - 1]
                o = BASE + self.nd + FEAT_C[lvl - 1]
                # up_flow feeds the next level's concat (T) and its warp (fp32)
                L[f"deconv{lvl}"].run_dgrad_form(self.flowT[lvl], nxt.slice(o, 2), y32=self.upflow32[lvl - 1], bias=True)
                L[f"upfeat{lvl}"].run_dgrad_form(X, nxt.slice(o + 2, 2), bias=True)
        # context network on the level-2 features, residual on flow2 (PWCNet.py:269-270)
        src = self.x[2]
        for i in range(6):
            L[f"dc_conv{i + 1}"].run_fwd_form(src, self.dc[i], slope=SLOPE)
            src = self.dc[i]
        L["dc_conv7"].run_fwd_form(src, self.dc7)
        f2 = self.flow32[2]
        _lib.call("mireg_cast_to_f32", f2.ptr, f2.ld, self.dc7.ptr, self.dc7.ld, f2.rows, 2, 1.0, 1.0, code, st)
        cast_from_f32(self.flowT[2], f2)
        L["deconv2"].run_dgrad_form(self.flowT[2], self.flowT[1], y32=self.flow32[1], bias=True)
        L["deconv1"].run_dgrad_form(self.flowT[1], None, y32=self.flow32[0], bias=True)
        return [self.flow32[l].nchw() for l in range(0, 7)]

    # ------------------------------------------------------------------------------------------------
    def _ensure_grad_buffers(self) -> None:
        if self.grads_ready:
            return
        new, hs, B = self.ws.new, self.hs, self.B
        z = lambda v: new(v.B, v.H, v.W, v.C)
        self.dx = {lvl: z(v) for lvl, v in self.x.items()}
        self.dpyr = {k: z(v) for k, v in self.pyr.items()}
        self.dwarped = {lvl: z(v) for lvl, v in self.warped.items()}
        self.dflowT = {lvl: new(B, *hs[lvl], 2) for lvl in range(0, 7)}
        self.dupflowT = {lvl: new(B, *hs[lvl], 2) for lvl in (5, 4, 3, 2)}
        self.dupfeatT = {lvl: new(B, *hs[lvl], 2) for lvl in (5, 4, 3, 2)}
        self.dupflow32 = {lvl: new(B, *hs[lvl], 2, dtype=F32, pad=2) for lvl in (5, 4, 3, 2)}
        self.dx32 = {lvl: new(B, *hs[lvl], FEAT_C[lvl], dtype=F32) for lvl in (5, 4, 3, 2)}      # overwritten by the warp backward
        self.warp_ws = {lvl: WarpBwdWorkspace(B * hs[lvl][0] * hs[lvl][1], self.ws.device) for lvl in (5, 4, 3, 2)}
        self.ddc = [z(v) for v in self.dc]
        self._zero_list = ([v.buf for v in self.dx.values()] + [v.buf for v in self.dpyr.values()] +
                           [v.buf for v in self.dwarped.values()] + [v.buf for v in self.dupflow32.values()] +
                           [v.buf for v in self.ddc])
        self.grads_ready = True

    def _conv_bwd(self, name: str, src: View, out: View, dout: View, dsrc, slot: int = 0, act: bool = True) -> None:
        """conv (+bias) (+LeakyReLU) backward; every gradient buffer was zeroed, so dgrad always accumulates."""
        lay = self.layers[name]
        if act:
            lrelu_bwd(dout, out, SLOPE, self.ws)
        ready = self.mark()
        lay.run_bias_grad(dout, accumulate=slot > 0)
        if dsrc is not None:
            lay.run_dgrad_form(dout, dsrc, accumulate=True)
        self.wgrad_async(lay, src, dout, slot, after=ready)          # critical-chain kernels first (graph node order)

    def _deconv_bwd(self, name: str, g_fine: View, x_coarse: View, dx_coarse: View) -> None:
        """ConvTranspose2d(k4,s2,p1) backward: g_fine = grad wrt its (2x larger) output."""
        lay = self.layers[name]
        ready = self.mark()
        lay.run_bias_grad(g_fine)
        lay.run_fwd_form(g_fine, dx_coarse, bias=False, accumulate=True)
        self.wgrad_async(lay, g_fine, x_coarse, after=ready)

    phase_opt_single_gpu = False      # measured: the two extra wgrad-stream joins cost more (10.3 ms) than the overlap buys (10.1 ms)

    def backward_phases(self, gflows, joined: bool = True):
        """Backward cut where gradient buckets complete, in parameter order tail -> head:
        [dc_conv*, deconv1/2, level-2 estimator] | [level 3..6 estimators, their flow / feature upsamplers] | [siamese pyramid]."""
        self._ensure_grad_buffers()
        L, code, st, B = self.layers, self.ws.code, _stream(), self.B
        g = list(gflows) + [None] * (7 - len(gflows))
        if getattr(self, "_zero_tab", None) is None:
            self._zero_tab = zero_many_table(self._zero_list, self.ws.device)
        _lib.call("mireg_zero_many", self._zero_tab[0].data_ptr(), self._zero_tab[1], self._zero_tab[2], st)
        for lvl in range(0, 7):                                   # loss gradients of the seven flows
            if g[lvl] is None:
                zero_tensors([self.dflowT[lvl].buf])
            else:
                nchw_to_view(g[lvl].contiguous(), 0, 2, self.dflowT[lvl])
        feat = lambda lvl, s: self.pyr[(lvl, s, 2)]
        dfeat = lambda lvl, s: self.dpyr[(lvl, s, 2)]

        def level(lvl):
            X, dX = self.x[lvl], self.dx[lvl]
            # predict_flow{lvl}: dflowT[lvl] now holds loss grad + everything pushed up from the finer level
            self._conv_bwd(f"predict_flow{lvl}", X, self.flowT[lvl], self.dflowT[lvl], dX, act=False)
            for j in range(4, -1, -1):                            # DenseNet estimator, last conv first
                off, cout = DENSE_OFF[j], DENSE[j]
                self._conv_bwd(f"conv{lvl}_{j}", X.slice(off + cout, X.C - off - cout), X.slice(off, cout),
                               dX.slice(off, cout), dX.slice(off + cout, X.C - off - cout))
            gcorr = dX.slice(BASE, self.nd)
            lrelu_bwd(gcorr, X.slice(BASE, self.nd), SLOPE, self.ws)
            c1, c2 = feat(lvl, "a"), feat(lvl, "b")
            cp = (FEAT_C[lvl] + 7) // 8 * 8
            if lvl == 6:
                d1, d2 = dfeat(6, "a"), dfeat(6, "b")
                correlation_bwd_views(gcorr, c1, c2, d1, d2, cp, FEAT_C[6], self.md, 1, 1, 1, code)
                return
            wv, dw, d1 = self.warped[lvl], self.dwarped[lvl], dfeat(lvl, "a")
            correlation_bwd_views(gcorr, c1, wv, d1, dw, cp, FEAT_C[lvl], self.md, 1, 1, 1, code)
            # c1 also sits in the concat
            gc1 = dX.slice(BASE + self.nd, FEAT_C[lvl])
            _lib.call("mireg_copy_channels", gc1.ptr, gc1.ld, d1.ptr, d1.ld, d1.rows, FEAT_C[lvl], 1, code, st)
            # warp backward -> d c2 (fp32 scatter) and d up_flow (fp32)
            dx32, duf32, d2 = self.dx32[lvl], self.dupflow32[lvl], dfeat(lvl, "b")
            pwc_warp_bwd_views(c2, self.upflow32[lvl], FLOW_SCALE[lvl], dw, dx32, duf32, FEAT_C[lvl], code, self.warp_ws[lvl])
            cast_from_f32(d2.slice(0, FEAT_C[lvl]), dx32.slice(0, FEAT_C[lvl]), 1.0, 1.0)
            # up_flow / up_feat gradients: aligned 2-channel staging buffers (the concat slices sit at odd offsets)
            o = BASE + self.nd + FEAT_C[lvl]
            guf, gft = self.dupflowT[lvl], self.dupfeatT[lvl]
            _lib.call("mireg_copy_channels", dX.slice(o, 2).ptr, dX.ld, guf.ptr, guf.ld, guf.rows, 2, 0, code, st)
            cast_from_f32(guf, duf32, 1.0, 1.0)
            _lib.call("mireg_copy_channels", dX.slice(o + 2, 2).ptr, dX.ld, gft.ptr, gft.ld, gft.rows, 2, 0, code, st)
            self._deconv_bwd(f"deconv{lvl + 1}", guf, self.flowT[lvl + 1], self.dflowT[lvl + 1])
            self._deconv_bwd(f"upfeat{lvl + 1}", gft, self.x[lvl + 1], self.dx[lvl + 1])

        def fine():
            # flow0 = deconv1(flow1), flow1 = deconv2(flow2)
            self._deconv_bwd("deconv1", self.dflowT[0], self.flowT[1], self.dflowT[1])
            self._deconv_bwd("deconv2", self.dflowT[1], self.flowT[2], self.dflowT[2])
            # flow2 = predict_flow2(x2) + dc_conv7(dc_conv6(...dc_conv1(x2)))
            dX2 = self.dx[2]
            self._conv_bwd("dc_conv7", self.dc[5], self.dc7, self.dflowT[2], self.ddc[5], act=False)
            for i in range(6, 0, -1):
                src, dsrc = (self.dc[i - 2], self.ddc[i - 2]) if i > 1 else (self.x[2], dX2)
                self._conv_bwd(f"dc_conv{i}", src, self.dc[i - 1], self.ddc[i - 1], dsrc)
            level(2)
            if joined:
                self.join_side()
                self.unpack_grads(self.PHASES[0])

        def coarse():
            for lvl in (3, 4, 5, 6):
                level(lvl)
            if joined:
                self.join_side()
                self.unpack_grads(self.PHASES[1])

        def pyramid():
            # siamese pyramid, coarse -> fine, one wgrad slot per stream
            for slot, s_ in enumerate("ab"):
                for lvl in range(6, 0, -1):
                    for i in (2, 1, 0):
                        if i > 0:
                            src, dsrc = self.pyr[(lvl, s_, i - 1)], self.dpyr[(lvl, s_, i - 1)]
                        elif lvl > 1:
                            src, dsrc = self.pyr[(lvl - 1, s_, 2)], self.dpyr[(lvl - 1, s_, 2)]
                        else:
                            src, dsrc = self.img[s_], None
                        self._conv_bwd(PYR_NAMES[lvl][i], src, self.pyr[(lvl, s_, i)], self.dpyr[(lvl, s_, i)], dsrc, slot)
            self.join_side()
            self.unpack_grads(self.PHASES[2] if joined else None)
        return [fine, coarse, pyramid]

    PHASES = (tuple([f"dc_conv{i}" for i in range(1, 8)] + ["deconv1", "deconv2", "predict_flow2"] + [f"conv2_{j}" for j in range(5)]),
              tuple(n for lvl in (3, 4, 5, 6) for n in [f"conv{lvl}_{j}" for j in range(5)] + [f"predict_flow{lvl}", f"deconv{lvl}", f"upfeat{lvl}"]),
              tuple(n for lvl in range(1, 7) for n in PYR_NAMES[lvl]))

    def phase_layers(self):
        return [(self.PHASES[0], ()), (self.PHASES[1], ()), (self.PHASES[2], ())]

    def phase_ranges(self):
        return [self.flat_range(self.PHASES[0], (), self._unused_tail), self.flat_range(self.PHASES[1]), self.flat_range(self.PHASES[2])]

    def backward(self, gflows) -> None:
        """gflows: gradients wrt (flow0 .. flow6) as (B,2,h,w) fp32 or None."""
        for phase in self.backward_phases(gflows, joined=False):   # one join + one reduce at the end
            phase()


class _PWCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module.engine_for(x)
        ctx.eng, ctx.module = eng, module
        return tuple(eng.forward(x, module.training))

    @staticmethod
    def backward(ctx, *g):
        ctx.eng.autograd_backward(g, ctx.module._findex)
        table = ctx.eng.param_grads()
        grads = grads_for_autograd(ctx.module.parameters(), table)
        return (None, None) + grads


class PWCDCNet(nn.Module, PackedOptimizerHook):
    """Drop-in for PWC.models.PWCNet.PWCDCNet(md=4)."""

    def __init__(self, md: int = 4, precision: str = "bf16"):
        super().__init__()
        self.md, self.precision = md, precision
        for lvl, (cin, cout) in enumerate(PYRAMID, start=1):
            n0, n1, n2 = PYR_NAMES[lvl]
            setattr(self, n0, _conv(cin, cout, 3, 2))
            setattr(self, n1, _conv(cout, cout, 3, 1))
            setattr(self, n2, _conv(cout, cout, 3, 1))
        self.corr = Correlation(pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=1, corr_multiply=1)
        self.leakyRELU = nn.LeakyReLU(SLOPE)
        nd = (2 * md + 1) ** 2
        dd = np.cumsum(DENSE)
        for lvl in (6, 5, 4, 3, 2):
            od = nd if lvl == 6 else nd + FEAT_C[lvl] + 4
            cin = od
            for j, cout in enumerate(DENSE):
                setattr(self, f"conv{lvl}_{j}", _conv(cin, cout))
                cin = od + int(dd[j])
            setattr(self, f"predict_flow{lvl}", nn.Conv2d(cin, 2, 3, 1, 1, bias=True))
            if lvl > 2:
                setattr(self, f"deconv{lvl}", nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=True))
                setattr(self, f"upfeat{lvl}", nn.ConvTranspose2d(cin, 2, 4, 2, 1, bias=True))
        self.deconv2 = nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=True)
        self.deconv1 = nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=True)
        self.deconv0 = nn.ConvTranspose2d(2, 2, 4, 4, 0, bias=True)          # defined, unused (PWCNet.py:126,274)
        cin = nd + 32 + 4 + int(dd[4])
        for i, (cout, d) in enumerate(DC, start=1):
            setattr(self, f"dc_conv{i}", _conv(cin, cout, 3, 1, d, d))
            cin = cout
        self.dc_conv7 = nn.Conv2d(32, 2, 3, 1, 1, bias=True)
        for m in self.modules():  # PWCNet.py:136-140
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight.data, mode="fan_in")
                if m.bias is not None:
                    m.bias.data.zero_()
        self._engines: Dict[tuple, PWCEngine] = {}

    def engine_for(self, x: torch.Tensor) -> PWCEngine:
        if not x.is_cuda:
            raise RuntimeError("mireg.PWCDCNet runs on the MI355X only; there is no CPU fallback")
        dtype = torch.bfloat16 if self.precision == "bf16" else torch.float32
        key = (tuple(x.shape), x.device, dtype, next(self.parameters()).data_ptr())
        if key not in self._engines:
            drop_engines(self)
            B, C, H, W = x.shape
            if C != 2:
                raise RuntimeError(f"PWCDCNet expects (B,2,H,W) [fixed, moving], got {tuple(x.shape)}")
            self._engines[key] = PWCEngine(self, B, H, W, x.device, dtype)
        return self._engines[key]

    def warp(self, x: torch.Tensor, flo: torch.Tensor) -> torch.Tensor:
        """Drop-in for PWCDCNet.warp (PWC/models/PWCNet.py:143-179): x (B,C,H,W), flo (B,2,H,W) -> warped x * mask,
        on the HIP kernel (values only; inside forward() the same kernel runs on the engine's NHWC buffers with its
        backward)."""
        if not x.is_cuda:
            raise RuntimeError("mireg.PWCDCNet.warp runs on the MI355X only; there is no CPU fallback")
        from .correlation import pwc_warp_views
        from .engine import Workspace
        B, C, H, W = x.shape
        ws = Workspace(x.device, torch.float32)
        xv, ov = ws.new(B, H, W, C), ws.new(B, H, W, C)
        xv.buf[..., :C] = x.detach().float().permute(0, 2, 3, 1)
        fv = ws.new(B, H, W, 2, dtype=torch.float32, pad=2)
        fv.buf[...] = flo.detach().float().permute(0, 2, 3, 1)
        pwc_warp_views(xv, fv, 1.0, ov, ws.code)
        return ov.nchw()[:, :C].contiguous()

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return tuple(_PWCFn.apply(self, x.float(), *self.parameters()))
        return tuple(self.engine_for(x).forward(x.float(), self.training))
