"""The FlowNet2 stack (reference flownet2/models.py:30-191; networks/FlowNetS.py, FlowNetSD.py, FlowNetFusion.py,
submodules.py) on the HIP engine -- SURVEY section 8(f) rank 1.

Same class / submodule names as the reference, so `state_dict` keys interchange.  The five sub-networks run on the
contraction kernels of the hot path (ring / halo GEMMs, thin heads, tiny upsamplers, BatchNorm kernels, the MFMA cost
volume inside FlowNetC); the glue between them is `Resample2d` / `ChannelNorm` / `Upsample` from flownet2_ops.hip plus a
handful of torch elementwise ops on 1-9 channel images (a subtraction, a scale, the concats).

Training: every sub-network is ONE autograd function whose backward is a HIP backward pass (FlowNetC's own; the FlowNetS
blocks and FlowNetFusion also return the gradient of their input, which the chain needs because their inputs are built
from the upstream flows), so `loss.backward()` through `FlowNet2` / `opticalFlowReg("flownet2")` works like the
reference's.  Eager launches; the fused `RegistrationTrainer` step does not cover this predictor.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .engine import F32, BatchNormAct, lrelu_bwd, nchw_to_view, zero_tensors
from .flownet2_ops import ChannelNorm, Resample2d, Upsample
from .flownetc import FlowNetC
from .flownets import (drop_engines, grads_for_autograd, PackedOptimizerHook, DECONV, ENCODER, PREDICT, SLOPE, FlowNetDecoderMixin, FlowNetSEngine, PredictorEngineBase, conv_block,
                       count_bn_batches, install_bn_counter_hook)

# (name, cin, cout, stride), all 3x3 -- flownet2/networks/FlowNetSD.py:17-29
SD_ENCODER = [("conv0", 2, 64, 1), ("conv1", 64, 64, 2), ("conv1_1", 64, 128, 1), ("conv2", 128, 128, 2), ("conv2_1", 128, 128, 1),
              ("conv3", 128, 256, 2), ("conv3_1", 256, 256, 1), ("conv4", 256, 512, 2), ("conv4_1", 512, 512, 1),
              ("conv5", 512, 512, 2), ("conv5_1", 512, 512, 1), ("conv6", 512, 1024, 2), ("conv6_1", 1024, 1024, 1)]
SD_INTER = {5: (1026, 512), 4: (770, 256), 3: (386, 128), 2: (194, 64)}          # FlowNetSD.py:36-39
SD_PREDICT = {6: 1024, 5: 512, 4: 256, 3: 128, 2: 64}                            # FlowNetSD.py:41-45


def i_conv(bn: bool, cin: int, cout: int) -> nn.Sequential:
    """flownet2/networks/submodules.py:20-30: conv (bias on) [+ BatchNorm], no activation."""
    layers: List[nn.Module] = [nn.Conv2d(cin, cout, 3, 1, 1, bias=True)]
    if bn:
        layers.append(nn.BatchNorm2d(cout))
    return nn.Sequential(*layers)


def deconv(cin: int, cout: int) -> nn.Sequential:
    """flownet2/networks/submodules.py:35-39."""
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True), nn.LeakyReLU(SLOPE, inplace=True))


def _xavier_(mod: nn.Module) -> None:
    """flownet2/networks/FlowNetSD.py:52-61 (the same loop closes every class of the stack)."""
    for m in mod.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if m.bias is not None:
                nn.init.uniform_(m.bias)
            nn.init.xavier_uniform_(m.weight)


class _StackEngine(PredictorEngineBase, FlowNetDecoderMixin):
    """Engine helpers shared by FlowNetSD and FlowNetFusion: conv (+BatchNorm) + activation blocks over NHWC views, forward and
    backward (`chain_backward`, the wgrad side stream and the gradient bookkeeping come from the FlowNetS machinery)."""

    def block(self, name: str, src, dst, training: bool, slope: float = SLOPE) -> None:
        lay = self.layers[name]
        if name in self.bns:
            lay.run_fwd_form(src, self.raw[name])
            self.bns[name].forward(self.raw[name], dst, training)
        else:
            lay.run_fwd_form(src, dst, slope=slope)

    def add_block(self, name: str, seq: nn.Sequential, stride: int, out_hw, slope: float = SLOPE) -> None:
        conv = seq[0]
        self.add_conv(name, conv, stride, 1)
        if len(seq) > 1 and isinstance(seq[1], nn.BatchNorm2d):
            self.bns[name] = BatchNormAct(seq[1], self.ws, slope)
            self.raw[name] = self.ws.new(self.B, *out_hw, conv.out_channels)

    # ---- backward pieces --------------------------------------------------------------------------------------------
    def load_flow_grad(self, g: Optional[torch.Tensor], dst) -> None:
        if g is None:
            zero_tensors([dst.buf])
        else:
            nchw_to_view(g.float().contiguous(), 0, 2, dst)

    def head_backward(self, name: str, feat, dflow, dfeat) -> None:
        """predict_flow: bias gradient, backward-data into dfeat (overwrites), backward-weights on the side stream."""
        pf = self.layers[name]
        ready = self.mark()
        pf.run_bias_grad(dflow)
        pf.run_dgrad_form(dflow, dfeat)
        self.wgrad_async(pf, feat, dflow, after=ready)

    def inter_backward(self, name: str, src, dinter, dsrc) -> None:
        """i_conv (conv with bias [+ BatchNorm], no activation): dinter = gradient of its output; dsrc is overwritten."""
        lay = self.layers[name]
        if name in self.bns:
            self.bns[name].backward(self.raw[name], dinter, self.draw[name])
            dy = self.draw[name]
        else:
            dy = dinter
        ready = self.mark()
        lay.run_bias_grad(dy)
        lay.run_dgrad_form(dy, dsrc)
        self.wgrad_async(lay, src, dy, after=ready)

    def up_backward(self, name: str, gup, flow_coarse, dflow_coarse, accumulate: bool) -> None:
        """ConvTranspose2d(2, 2, 4, 2, 1) flow upsampler: gup = gradient of its (fine) output slice."""
        up = self.layers[name]
        ready = self.mark()
        up.run_bias_grad(gup)
        up.run_fwd_form(gup, dflow_coarse, bias=False, accumulate=accumulate)
        self.wgrad_async(up, gup, flow_coarse, after=ready)

    def deconv_backward(self, name: str, gde, act_slice, src, dsrc) -> None:
        """deconv + LeakyReLU: gde = gradient of its activated output slice (masked in place); accumulates into dsrc."""
        de = self.layers[name]
        lrelu_bwd(gde, act_slice, SLOPE, self.ws)
        ready = self.mark()
        de.run_bias_grad(gde)
        de.run_fwd_form(gde, dsrc, bias=False, accumulate=True)
        self.wgrad_async(de, gde, src, after=ready)

    def finish_backward(self) -> None:
        self.join_side()
        self.unpack_grads()


class FlowNetSDEngine(_StackEngine):
    def __init__(self, module: "FlowNetSD", B: int, H: int, W: int, device, dtype: torch.dtype):
        super().__init__(module, B, H, W, device, dtype)
        if H % 64 or W % 64:
            raise RuntimeError(f"FlowNetSD engine needs H, W divisible by 64, got {H}x{W}")
        ws, m, new = self.ws, module, self.ws.new
        hs = {lvl: (H >> lvl, W >> lvl) for lvl in range(0, 7)}
        self.hs, self.raw, self.bn, self.grads_ready = hs, {}, m.batchNorm, False
        lvl = 0
        for name, cin, cout, s in SD_ENCODER:
            lvl += s - 1
            self.add_block(name, getattr(m, name), s, hs[lvl])
        for l, (cin, cout) in DECONV.items():
            self.add_conv(f"deconv{l}", getattr(m, f"deconv{l}")[0], 2, 1)
        for l in SD_INTER:
            self.add_block(f"inter_conv{l}", getattr(m, f"inter_conv{l}"), 1, hs[l], slope=1.0)
        for l in SD_PREDICT:
            self.add_conv(f"predict_flow{l}", getattr(m, f"predict_flow{l}"), 1, 1)
        for l in (6, 5, 4, 3):
            self.add_conv(f"up{l}", getattr(m, f"upsampled_flow{l}_to_{l - 1}"), 2, 1)
        self.x8 = new(B, H, W, 2)
        self.a0, self.a1, self.a11 = new(B, *hs[0], 64), new(B, *hs[1], 64), new(B, *hs[1], 128)
        self.a2 = new(B, *hs[2], 128)
        self.cat = {2: new(B, *hs[2], 194), 3: new(B, *hs[3], 386), 4: new(B, *hs[4], 770), 5: new(B, *hs[5], 1026)}
        self.skip_c = {2: 128, 3: 256, 4: 512, 5: 512}
        self.a3, self.a4, self.a5 = new(B, *hs[3], 256), new(B, *hs[4], 512), new(B, *hs[5], 512)
        self.a6, self.a61 = new(B, *hs[6], 1024), new(B, *hs[6], 1024)
        self.inter = {l: new(B, *hs[l], cout) for l, (_, cout) in SD_INTER.items()}
        self.flow32 = {l: new(B, *hs[l], 2, dtype=F32, pad=2) for l in SD_PREDICT}
        self.flowT = {l: new(B, *hs[l], 2) for l in SD_PREDICT}

    def forward(self, x: torch.Tensor, training: bool) -> List[torch.Tensor]:
        L, c = self.layers, self.cat
        self.pack_weights()
        nchw_to_view(x.contiguous(), 0, 2, self.x8)
        self.block("conv0", self.x8, self.a0, training)
        self.block("conv1", self.a0, self.a1, training)
        self.block("conv1_1", self.a1, self.a11, training)
        self.block("conv2", self.a11, self.a2, training)
        self.block("conv2_1", self.a2, c[2].slice(0, 128), training)
        self.block("conv3", c[2].slice(0, 128), self.a3, training)
        self.block("conv3_1", self.a3, c[3].slice(0, 256), training)
        self.block("conv4", c[3].slice(0, 256), self.a4, training)
        self.block("conv4_1", self.a4, c[4].slice(0, 512), training)
        self.block("conv5", c[4].slice(0, 512), self.a5, training)
        self.block("conv5_1", self.a5, c[5].slice(0, 512), training)
        self.block("conv6", c[5].slice(0, 512), self.a6, training)
        self.block("conv6_1", self.a6, self.a61, training)
        feat = self.a61
        L["predict_flow6"].run_fwd_form(feat, self.flowT[6], y32=self.flow32[6])
        for lvl in (5, 4, 3, 2):
            cs, cd = self.skip_c[lvl], DECONV[lvl][1]
            L[f"up{lvl + 1}"].run_dgrad_form(self.flowT[lvl + 1], c[lvl].slice(cs + cd, 2), bias=True)
            L[f"deconv{lvl}"].run_dgrad_form(feat, c[lvl].slice(cs, cd), slope=SLOPE, bias=True)
            feat = c[lvl]
            self.block(f"inter_conv{lvl}", feat, self.inter[lvl], training, slope=1.0)
            L[f"predict_flow{lvl}"].run_fwd_form(self.inter[lvl], self.flowT[lvl], y32=self.flow32[lvl])
        flows = [self.flow32[2].nchw()]
        if training:
            flows += [self.flow32[l].nchw() for l in (3, 4, 5, 6)]
        return flows

    def _ensure_grad_buffers(self) -> None:
        if self.grads_ready:
            return
        new, hs, B = self.ws.new, self.hs, self.B
        self.dcat = {l: new(B, *hs[l], v.C) for l, v in self.cat.items()}
        self.dinter = {l: new(B, *hs[l], v.C) for l, v in self.inter.items()}
        self.dflowT = {l: new(B, *hs[l], 2) for l in SD_PREDICT}
        self.da61, self.da6 = new(B, *hs[6], 1024), new(B, *hs[6], 1024)
        self.da5, self.da4, self.da3 = new(B, *hs[5], 512), new(B, *hs[4], 512), new(B, *hs[3], 256)
        self.da2, self.da11, self.da1, self.da0 = new(B, *hs[2], 128), new(B, *hs[1], 128), new(B, *hs[1], 64), new(B, *hs[0], 64)
        self.draw = {n: new(B, v.H, v.W, v.C) for n, v in self.raw.items()}
        self.grads_ready = True

    def backward(self, gflows) -> None:
        """gflows: gradients wrt (flow2, flow3, flow4, flow5, flow6) [training arity] or (flow2,), (B,2,h,w) fp32 or None."""
        self._ensure_grad_buffers()
        g = list(gflows) + [None] * (5 - len(gflows))
        glvl = {2: g[0], 3: g[1], 4: g[2], 5: g[3], 6: g[4]}
        c, dc = self.cat, self.dcat
        # decoder, fine to coarse (FlowNetSD.py:77-100 backwards): dcat[l] is complete when level l is entered
        self.load_flow_grad(glvl[2], self.dflowT[2])
        self.head_backward("predict_flow2", self.inter[2], self.dflowT[2], self.dinter[2])
        self.inter_backward("inter_conv2", c[2], self.dinter[2], dc[2])
        for lvl in (2, 3, 4, 5):
            cs, cd, up = self.skip_c[lvl], DECONV[lvl][1], lvl + 1
            self.load_flow_grad(glvl[up], self.dflowT[up])
            self.up_backward(f"up{up}", dc[lvl].slice(cs + cd, 2), self.flowT[up], self.dflowT[up], accumulate=True)
            if up == 6:
                src, dsrc = self.a61, self.da61
                self.head_backward("predict_flow6", src, self.dflowT[6], dsrc)
            else:
                src, dsrc = c[up], dc[up]
                self.head_backward(f"predict_flow{up}", self.inter[up], self.dflowT[up], self.dinter[up])
                self.inter_backward(f"inter_conv{up}", src, self.dinter[up], dsrc)
            self.deconv_backward(f"deconv{lvl}", dc[lvl].slice(cs, cd), c[lvl].slice(cs, cd), src, dsrc)
        # encoder, deep to shallow; the concat slices already hold the decoder's contributions
        s2, s3, s4, s5 = c[2].slice(0, 128), c[3].slice(0, 256), c[4].slice(0, 512), c[5].slice(0, 512)
        d2, d3, d4, d5 = dc[2].slice(0, 128), dc[3].slice(0, 256), dc[4].slice(0, 512), dc[5].slice(0, 512)
        for name, src, dst, dsrc, acc, ddst in (
                ("conv6_1", self.a6, self.a61, self.da6, False, self.da61), ("conv6", s5, self.a6, d5, True, self.da6),
                ("conv5_1", self.a5, s5, self.da5, False, d5), ("conv5", s4, self.a5, d4, True, self.da5),
                ("conv4_1", self.a4, s4, self.da4, False, d4), ("conv4", s3, self.a4, d3, True, self.da4),
                ("conv3_1", self.a3, s3, self.da3, False, d3), ("conv3", s2, self.a3, d2, True, self.da3),
                ("conv2_1", self.a2, s2, self.da2, False, d2), ("conv2", self.a11, self.a2, self.da11, False, self.da2),
                ("conv1_1", self.a1, self.a11, self.da1, False, self.da11), ("conv1", self.a0, self.a1, self.da0, False, self.da1),
                ("conv0", self.x8, self.a0, None, False, self.da0)):
            self.chain_backward(name, src, dst, dsrc, acc, ddst)
        self.finish_backward()


class FlowNetFusionEngine(_StackEngine):
    def __init__(self, module: "FlowNetFusion", B: int, H: int, W: int, device, dtype: torch.dtype):
        super().__init__(module, B, H, W, device, dtype)
        if H % 4 or W % 4:
            raise RuntimeError(f"FlowNetFusion engine needs H, W divisible by 4, got {H}x{W}")
        m, new = module, self.ws.new
        hs = {lvl: (H >> lvl, W >> lvl) for lvl in range(0, 3)}
        self.hs, self.raw, self.bn, self.grads_ready = hs, {}, m.batchNorm, False
        for name, s, lvl in (("conv0", 1, 0), ("conv1", 2, 1), ("conv1_1", 1, 1), ("conv2", 2, 2), ("conv2_1", 1, 2)):
            self.add_block(name, getattr(m, name), s, hs[lvl])
        self.add_conv("deconv1", m.deconv1[0], 2, 1)
        self.add_conv("deconv0", m.deconv0[0], 2, 1)
        self.add_block("inter_conv1", m.inter_conv1, 1, hs[1], slope=1.0)
        self.add_block("inter_conv0", m.inter_conv0, 1, hs[0], slope=1.0)
        for l in (2, 1, 0):
            self.add_conv(f"predict_flow{l}", getattr(m, f"predict_flow{l}"), 1, 1)
        self.add_conv("up2", m.upsampled_flow2_to_1, 2, 1)
        self.add_conv("up1", m.upsampled_flow1_to_0, 2, 1)
        self.x9 = new(B, H, W, 9)
        self.cat0, self.cat1 = new(B, *hs[0], 82), new(B, *hs[1], 162)       # [conv0 64 | deconv0 16 | up 2], [conv1_1 128 | deconv1 32 | up 2]
        self.a1, self.a2, self.a21 = new(B, *hs[1], 64), new(B, *hs[2], 128), new(B, *hs[2], 128)
        self.i1, self.i0 = new(B, *hs[1], 32), new(B, *hs[0], 16)
        self.flow32 = {l: new(B, *hs[l], 2, dtype=F32, pad=2) for l in (2, 1, 0)}
        self.flowT = {l: new(B, *hs[l], 2) for l in (2, 1, 0)}

    def forward(self, x: torch.Tensor, training: bool) -> torch.Tensor:
        L = self.layers
        self.pack_weights()
        nchw_to_view(x.contiguous(), 0, 9, self.x9)
        self.block("conv0", self.x9, self.cat0.slice(0, 64), training)
        self.block("conv1", self.cat0.slice(0, 64), self.a1, training)
        self.block("conv1_1", self.a1, self.cat1.slice(0, 128), training)
        self.block("conv2", self.cat1.slice(0, 128), self.a2, training)
        self.block("conv2_1", self.a2, self.a21, training)
        L["predict_flow2"].run_fwd_form(self.a21, self.flowT[2], y32=self.flow32[2])
        L["up2"].run_dgrad_form(self.flowT[2], self.cat1.slice(160, 2), bias=True)
        L["deconv1"].run_dgrad_form(self.a21, self.cat1.slice(128, 32), slope=SLOPE, bias=True)
        self.block("inter_conv1", self.cat1, self.i1, training, slope=1.0)
        L["predict_flow1"].run_fwd_form(self.i1, self.flowT[1], y32=self.flow32[1])
        L["up1"].run_dgrad_form(self.flowT[1], self.cat0.slice(80, 2), bias=True)
        L["deconv0"].run_dgrad_form(self.cat1, self.cat0.slice(64, 16), slope=SLOPE, bias=True)
        self.block("inter_conv0", self.cat0, self.i0, training, slope=1.0)
        L["predict_flow0"].run_fwd_form(self.i0, self.flowT[0], y32=self.flow32[0])
        return self.flow32[0].nchw()

    def _ensure_grad_buffers(self) -> None:
        if self.grads_ready:
            return
        new, hs, B = self.ws.new, self.hs, self.B
        self.dcat0, self.dcat1 = new(B, *hs[0], 82), new(B, *hs[1], 162)
        self.da21, self.da2, self.da1 = new(B, *hs[2], 128), new(B, *hs[2], 128), new(B, *hs[1], 64)
        self.di1, self.di0 = new(B, *hs[1], 32), new(B, *hs[0], 16)
        self.dflowT = {l: new(B, *hs[l], 2) for l in (2, 1, 0)}
        self.dx9 = new(B, *hs[0], 9)
        self.draw = {n: new(B, v.H, v.W, v.C) for n, v in self.raw.items()}
        self.grads_ready = True

    def backward(self, gflows) -> None:
        """gflows: (gradient wrt flow0,) as (B,2,H,W) fp32.  Leaves d loss / d input in dx9 (FlowNetFusion.py:43-66 backwards)."""
        self._ensure_grad_buffers()
        c0, c1, d0, d1 = self.cat0, self.cat1, self.dcat0, self.dcat1
        self.load_flow_grad(gflows[0], self.dflowT[0])
        self.head_backward("predict_flow0", self.i0, self.dflowT[0], self.di0)
        self.inter_backward("inter_conv0", c0, self.di0, d0)
        # level 0 -> 1
        self.up_backward("up1", d0.slice(80, 2), self.flowT[1], self.dflowT[1], accumulate=False)
        self.head_backward("predict_flow1", self.i1, self.dflowT[1], self.di1)
        self.inter_backward("inter_conv1", c1, self.di1, d1)
        self.deconv_backward("deconv0", d0.slice(64, 16), c0.slice(64, 16), c1, d1)
        # level 1 -> 2
        self.up_backward("up2", d1.slice(160, 2), self.flowT[2], self.dflowT[2], accumulate=False)
        self.head_backward("predict_flow2", self.a21, self.dflowT[2], self.da21)
        self.deconv_backward("deconv1", d1.slice(128, 32), c1.slice(128, 32), self.a21, self.da21)
        for name, src, dst, dsrc, acc, ddst in (
                ("conv2_1", self.a2, self.a21, self.da2, False, self.da21), ("conv2", c1.slice(0, 128), self.a2, d1.slice(0, 128), True, self.da2),
                ("conv1_1", self.a1, c1.slice(0, 128), self.da1, False, d1.slice(0, 128)),
                ("conv1", c0.slice(0, 64), self.a1, d0.slice(0, 64), True, self.da1),
                ("conv0", self.x9, c0.slice(0, 64), self.dx9, False, d0.slice(0, 64))):
            self.chain_backward(name, src, dst, dsrc, acc, ddst)
        self.finish_backward()

    def input_grad(self) -> torch.Tensor:
        return self.dx9.nchw().float()


class _StackFn(torch.autograd.Function):
    """One sub-network of the stack as one autograd node: forward = the engine's kernel sequence, backward = its HIP backward
    pass.  `skip` leading engine outputs are not returned (the FlowNetS engine's 256x256 top flow)."""

    @staticmethod
    def forward(ctx, module, cls, channels, skip, want_dx, x, *params):
        eng = module._engine(x, cls, channels)
        if want_dx:
            eng.want_dx = True
        out = eng.forward(x, module.training)
        eng.training_cache = module.training
        if module.training:
            count_bn_batches(eng)
        outs = list(out) if isinstance(out, (list, tuple)) else [out]
        ctx.eng, ctx.module, ctx.skip, ctx.dx = eng, module, skip, want_dx and x.requires_grad
        ctx.n_out = len(outs)
        return tuple(outs[skip:])

    @staticmethod
    def backward(ctx, *g):
        eng = ctx.eng
        g = (None,) * ctx.skip + tuple(g)
        if isinstance(eng, FlowNetSEngine):
            g = g + (None,) * (6 - len(g))
        eng.autograd_backward(g, ctx.module._findex)
        table = eng.param_grads()
        grads = grads_for_autograd(ctx.module.parameters(), table)
        dx = eng.input_grad() if ctx.dx else None
        return (None, None, None, None, None, dx) + grads


class _EngineCache(PackedOptimizerHook):
    """One engine per (input shape, device, precision, parameter storage)."""

    def _engine(self, x: torch.Tensor, cls, channels: int):
        if not x.is_cuda:
            raise RuntimeError(f"mireg.{type(self).__name__} runs on the MI355X only; there is no CPU fallback")
        dtype = torch.bfloat16 if self.precision == "bf16" else torch.float32
        p0 = next(self.parameters())
        if p0.device != x.device:
            raise RuntimeError("model and input are on different devices")
        key = (tuple(x.shape), x.device, dtype, p0.data_ptr())
        if key not in self._engines:
            drop_engines(self)
            B, C, H, W = x.shape
            if C != channels:
                raise RuntimeError(f"{type(self).__name__} expects (B,{channels},H,W), got {tuple(x.shape)}")
            self._engines[key] = cls(self, B, H, W, x.device, dtype)
            install_bn_counter_hook(self)
        return self._engines[key]


class FlowNet2S(nn.Module, _EngineCache):
    """flownet2.networks.FlowNetS.FlowNetS(args, input_channels=6, batchNorm=True): the FlowNetS blocks of the stack --
    biases on for the deconvolutions and flow heads, off for the flow upsamplers; returns (flow2, ...) without the
    registration model's 256x256 top flow.  Runs on FlowNetSEngine (the 6-channel 7x7 conv1 takes the generic GEMM)."""

    def __init__(self, args=None, input_channels: int = 6, batchNorm: bool = True, precision: str = "bf16"):
        super().__init__()
        self.batchNorm, self.precision, self.input_channels = batchNorm, precision, input_channels
        for name, cin, cout, k, s in ENCODER:
            setattr(self, name, conv_block(batchNorm, input_channels if name == "conv1" else cin, cout, k, s))
        for lvl, (cin, cout) in DECONV.items():
            setattr(self, f"deconv{lvl}", deconv(cin, cout))
        for lvl, cin in PREDICT.items():
            setattr(self, f"predict_flow{lvl}", nn.Conv2d(cin, 2, 3, 1, 1, bias=True))
        for lvl in (6, 5, 4, 3):
            setattr(self, f"upsampled_flow{lvl}_to_{lvl - 1}", nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=False))
        _xavier_(self)
        self.upsample1 = nn.Upsample(scale_factor=4, mode="bilinear")
        self._engines: Dict[tuple, FlowNetSEngine] = {}

    def forward(self, x):
        # the engine's first output is the 256x256 top flow of FlowNetS/FlowNetS.py:83, which this variant does not have
        return _StackFn.apply(self, FlowNetSEngine, self.input_channels, 1, True, x.float(), *self.parameters())


class FlowNetSD(nn.Module, _EngineCache):
    """Drop-in for flownet2.networks.FlowNetSD.FlowNetSD(args, batchNorm=True)."""

    def __init__(self, args=None, batchNorm: bool = True, precision: str = "bf16"):
        super().__init__()
        self.batchNorm, self.precision = batchNorm, precision
        for name, cin, cout, s in SD_ENCODER:
            setattr(self, name, conv_block(batchNorm, cin, cout, 3, s))
        for lvl, (cin, cout) in DECONV.items():
            setattr(self, f"deconv{lvl}", deconv(cin, cout))
        for lvl, (cin, cout) in SD_INTER.items():
            setattr(self, f"inter_conv{lvl}", i_conv(batchNorm, cin, cout))
        for lvl, cin in SD_PREDICT.items():
            setattr(self, f"predict_flow{lvl}", nn.Conv2d(cin, 2, 3, 1, 1, bias=True))
        for lvl in (6, 5, 4, 3):
            setattr(self, f"upsampled_flow{lvl}_to_{lvl - 1}", nn.ConvTranspose2d(2, 2, 4, 2, 1))
        _xavier_(self)
        self.upsample1 = nn.Upsample(scale_factor=4, mode="bilinear")
        self._engines: Dict[tuple, FlowNetSDEngine] = {}

    def forward(self, x):
        return _StackFn.apply(self, FlowNetSDEngine, 2, 0, False, x.float(), *self.parameters())


class FlowNetFusion(nn.Module, _EngineCache):
    """Drop-in for flownet2.networks.FlowNetFusion.FlowNetFusion(args, batchNorm=True): 9 channels in, full-resolution flow out."""

    def __init__(self, args=None, batchNorm: bool = True, precision: str = "bf16"):
        super().__init__()
        self.batchNorm, self.precision = batchNorm, precision
        self.conv0 = conv_block(batchNorm, 9, 64)
        self.conv1 = conv_block(batchNorm, 64, 64, 3, 2)
        self.conv1_1 = conv_block(batchNorm, 64, 128)
        self.conv2 = conv_block(batchNorm, 128, 128, 3, 2)
        self.conv2_1 = conv_block(batchNorm, 128, 128)
        self.deconv1 = deconv(128, 32)
        self.deconv0 = deconv(162, 16)
        self.inter_conv1 = i_conv(batchNorm, 162, 32)
        self.inter_conv0 = i_conv(batchNorm, 82, 16)
        self.predict_flow2 = nn.Conv2d(128, 2, 3, 1, 1, bias=True)
        self.predict_flow1 = nn.Conv2d(32, 2, 3, 1, 1, bias=True)
        self.predict_flow0 = nn.Conv2d(16, 2, 3, 1, 1, bias=True)
        self.upsampled_flow2_to_1 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        self.upsampled_flow1_to_0 = nn.ConvTranspose2d(2, 2, 4, 2, 1)
        _xavier_(self)
        self._engines: Dict[tuple, FlowNetFusionEngine] = {}

    def forward(self, x):
        return _StackFn.apply(self, FlowNetFusionEngine, 9, 0, True, x.float(), *self.parameters())[0]


class FlowNet2(nn.Module):
    """Drop-in for flownet2.models.FlowNet2(args, batchNorm, div_flow=20.) as the reference builds it (models.py:225,
    batchNorm=True; the rgb-mean normalisation is commented out in its forward, models.py:122-126)."""

    def __init__(self, args=None, batchNorm: bool = False, div_flow: float = 20.0, precision: str = "bf16"):
        super().__init__()
        self.batchNorm, self.div_flow, self.precision = batchNorm, div_flow, precision
        self.rgb_max = getattr(args, "rgb_max", 255.0)
        self.channelnorm = ChannelNorm()
        self.flownetc = FlowNetC(args, batchNorm=batchNorm, precision=precision)
        self.upsample1 = Upsample(scale_factor=4, mode="bilinear")
        self.resample1 = Resample2d()
        self.flownets_1 = FlowNet2S(args, batchNorm=batchNorm, precision=precision)
        self.upsample2 = Upsample(scale_factor=4, mode="bilinear")
        self.resample2 = Resample2d()
        self.flownets_2 = FlowNet2S(args, batchNorm=batchNorm, precision=precision)
        self.flownets_d = FlowNetSD(args, batchNorm=batchNorm, precision=precision)
        self.upsample3 = Upsample(scale_factor=4, mode="nearest")
        self.upsample4 = Upsample(scale_factor=4, mode="nearest")
        self.resample3 = Resample2d()
        self.resample4 = Resample2d()
        self.flownetfusion = FlowNetFusion(args, batchNorm=batchNorm, precision=precision)
        _xavier_(self)

    def stages(self, inputs: torch.Tensor):
        """(flownetc_flow2, flownets1_flow2, flownets2_flow2, flownetsd_flow2, fused flow) -- flownet2/models.py:128-189."""
        inputs = inputs.float()
        x1, x2 = inputs[:, 0:1].contiguous(), inputs[:, 1:2].contiguous()
        c2 = self.flownetc(inputs)[0].contiguous()            # engine outputs are permuted views of buffers the next forward reuses
        cflow = self.upsample1(c2 * self.div_flow)
        r1 = self.resample1(x2, cflow)
        cat1 = torch.cat((inputs, r1, cflow / self.div_flow, self.channelnorm(x1 - r1)), 1)
        s1 = self.flownets_1(cat1)[0].contiguous()
        s1flow = self.upsample2(s1 * self.div_flow)
        r2 = self.resample2(x2, s1flow)
        cat2 = torch.cat((inputs, r2, s1flow / self.div_flow, self.channelnorm(x1 - r2)), 1)
        s2 = self.flownets_2(cat2)[0].contiguous()
        s2flow = self.upsample4(s2 * self.div_flow)
        n_s2 = self.channelnorm(s2flow)
        d_s2 = self.channelnorm(x1 - self.resample4(x2, s2flow))
        sd = self.flownets_d(inputs)[0].contiguous()
        sdflow = self.upsample3(sd / self.div_flow)
        n_sd = self.channelnorm(sdflow)
        d_sd = self.channelnorm(x1 - self.resample3(x2, sdflow))
        cat3 = torch.cat((x1, sdflow, s2flow, n_sd, n_s2, d_sd, d_s2), 1)
        return c2, s1, s2, sd, self.flownetfusion(cat3).contiguous()

    def forward(self, inputs: torch.Tensor):
        fused = self.stages(inputs)[-1]
        return fused, fused
