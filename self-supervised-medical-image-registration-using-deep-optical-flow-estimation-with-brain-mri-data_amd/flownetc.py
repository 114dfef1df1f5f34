"""FlowNetC (reference flownet2/networks/FlowNetC.py:13-130, 1-channel streams) on the HIP engine.

Same submodule names / state_dict keys / train-eval arity as the reference: train -> (flow2..flow6),
eval -> (flow2,).  The siamese conv1-3 run once per stream (BatchNorm batch statistics per call, exactly
as the reference's two self.conv1(...) calls), the 441-channel cost volume comes from the MFMA correlation
kernel with its LeakyReLU fused, conv_redir and the cost volume are written straight into the 473-channel
input buffer of conv3_1 (no torch.cat), and the refinement decoder is shared with FlowNetS.
Backward: decoder shared with FlowNetS, cost-volume backward on the exact-fp32 MFMA (csrc/correlation.hip), the two
siamese streams accumulate into one weight gradient (one wgrad slab group per stream).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .correlation import correlation_bwd_views, Correlation, correlation_views
from .engine import BatchNormAct, lrelu_bwd, nchw_to_view
from .flownets import (drop_engines, grads_for_autograd, PackedOptimizerHook, DECONV, ENCODER, PREDICT, SLOPE, FlowNetDecoderMixin, PredictorEngineBase, conv_block, count_bn_batches,
                       install_bn_counter_hook)


class FlowNetCEngine(PredictorEngineBase, FlowNetDecoderMixin):
    def __init__(self, module: "FlowNetC", B: int, H: int, W: int, device, dtype: torch.dtype):
        super().__init__(module, B, H, W, device, dtype)
        if H % 64 or W % 64:
            raise RuntimeError(f"FlowNetC engine needs H, W divisible by 64, got {H}x{W}")
        ws, m = self.ws, module
        self.bn = m.batchNorm
        hs = {lvl: (H >> lvl, W >> lvl) for lvl in range(1, 7)}
        self.hs = hs
        spec = [("conv1", 7, 2), ("conv2", 5, 2), ("conv3", 5, 2), ("conv_redir", 1, 1), ("conv3_1", 3, 1)] + \
               [(n, k, s) for n, _, _, k, s in ENCODER[4:]]
        for name, k, s in spec:
            seq = getattr(m, name)
            self.add_conv(name, seq[0], s, (k - 1) // 2, uses=2 if name in ("conv1", "conv2", "conv3") else 1)
        # siamese layers keep one BatchNormAct per stream (separate batch statistics / saved scale-shift)
        for name in ("conv1", "conv2", "conv3"):
            if self.bn:
                self.bns[name + "@a"] = BatchNormAct(getattr(m, name)[1], ws, SLOPE)
                self.bns[name + "@b"] = BatchNormAct(getattr(m, name)[1], ws, SLOPE)
                self.bns[name + "@b"].grad_g = self.bns[name + "@a"].grad_g       # one gradient per shared parameter
                self.bns[name + "@b"].grad_b = self.bns[name + "@a"].grad_b
        for name, _, _ in spec[3:]:
            if self.bn:
                self.bns[name] = BatchNormAct(getattr(m, name)[1], ws, SLOPE)
        self.setup_decoder(m)
        new = ws.new
        self.xa, self.xb = new(B, H, W, 1), new(B, H, W, 1)
        self.c1 = {s: new(B, *hs[1], 64) for s in "ab"}
        self.cat = {2: new(B, *hs[2], 194), 3: new(B, *hs[3], 386), 4: new(B, *hs[4], 770), 5: new(B, *hs[5], 1026)}
        self.skip_c = {2: 128, 3: 256, 4: 512, 5: 512}
        self.c2b = new(B, *hs[2], 128)
        self.c3 = {s: new(B, *hs[3], 256) for s in "ab"}
        self.in31 = new(B, *hs[3], 473)                  # [conv_redir 32 | corr 441]
        self.a4, self.a5 = new(B, *hs[4], 512), new(B, *hs[5], 512)
        self.a6, self.a61 = new(B, *hs[6], 1024), new(B, *hs[6], 1024)
        shapes = {"conv1@a": (1, 64), "conv2@a": (2, 128), "conv3@a": (3, 256), "conv1@b": (1, 64), "conv2@b": (2, 128),
                  "conv3@b": (3, 256), "conv_redir": (3, 32), "conv3_1": (3, 256), "conv4": (4, 512), "conv4_1": (4, 512),
                  "conv5": (5, 512), "conv5_1": (5, 512), "conv6": (6, 1024), "conv6_1": (6, 1024)}
        self.raw = {n: new(B, *hs[l], c) for n, (l, c) in shapes.items()} if self.bn else {}
        self.grads_ready = False

    def _block(self, name: str, src, dst, training: bool, bn_key: str = None) -> None:
        lay = self.layers[name]
        key = bn_key or name
        if self.bn:
            lay.run_fwd_form(src, self.raw[key])
            self.bns[key].forward(self.raw[key], dst, training)
        else:
            lay.run_fwd_form(src, dst, slope=SLOPE)

    def forward(self, x: torch.Tensor, training: bool) -> List[torch.Tensor]:
        c = self.cat
        self.training_cache = training
        self.pack_weights()
        x = x.contiguous()
        nchw_to_view(x, 0, 1, self.xa)
        nchw_to_view(x, 1, 1, self.xb)
        # top stream (fixed image): conv2 output is the level-2 skip
        self._block("conv1", self.xa, self.c1["a"], training, "conv1@a")
        self._block("conv2", self.c1["a"], c[2].slice(0, 128), training, "conv2@a")
        self._block("conv3", c[2].slice(0, 128), self.c3["a"], training, "conv3@a")
        # bottom stream (moving image), same weights
        self._block("conv1", self.xb, self.c1["b"], training, "conv1@b")
        self._block("conv2", self.c1["b"], self.c2b, training, "conv2@b")
        self._block("conv3", self.c2b, self.c3["b"], training, "conv3@b")
        # merge: [conv_redir(c3a) | lrelu(corr(c3a, c3b))] -> conv3_1
        correlation_views(self.c3["a"], self.c3["b"], self.in31.slice(32, 441), 256, 20, 2, SLOPE, self.ws.code)
        self._block("conv_redir", self.c3["a"], self.in31.slice(0, 32), training)
        self._block("conv3_1", self.in31, c[3].slice(0, 256), training)
        self._block("conv4", c[3].slice(0, 256), self.a4, training)
        self._block("conv4_1", self.a4, c[4].slice(0, 512), training)
        self._block("conv5", c[4].slice(0, 512), self.a5, training)
        self._block("conv5_1", self.a5, c[5].slice(0, 512), training)
        self._block("conv6", c[5].slice(0, 512), self.a6, training)
        self._block("conv6_1", self.a6, self.a61, training)
        self.decoder_forward()
        flows = [self.flow32[2].nchw()]
        if training:
            flows += [self.flow32[l].nchw() for l in (3, 4, 5, 6)]
        return flows

    def _ensure_grad_buffers(self) -> None:
        if self.grads_ready:
            return
        new, hs, B = self.ws.new, self.hs, self.B
        self.setup_decoder_grads()
        self.da4, self.da5, self.da6 = new(B, *hs[4], 512), new(B, *hs[5], 512), new(B, *hs[6], 1024)
        self.din31 = new(B, *hs[3], 473)
        self.dc3 = {s: new(B, *hs[3], 256) for s in "ab"}
        self.dc2b = new(B, *hs[2], 128)
        self.dc1 = {s: new(B, *hs[1], 64) for s in "ab"}
        self.draw = {n: new(B, v.H, v.W, v.C) for n, v in self.raw.items()}
        self.grads_ready = True

    DEC_LAYERS = [f"deconv{l}" for l in DECONV] + [f"predict_flow{l}" for l in PREDICT] + [f"up{l}" for l in (6, 5, 4, 3)]
    PHASE_ENC = (("conv6_1", "conv6", "conv5_1", "conv5", "conv4_1", "conv4"), ("conv3_1", "conv_redir", "conv3", "conv2", "conv1"))

    def _bn_names(self, names) -> tuple:
        if not self.bn:
            return ()
        return tuple(k for n in names for k in ((n + "@a", n + "@b") if n in ("conv1", "conv2", "conv3") else (n,)))

    def phase_layers(self):
        return [(tuple(self.DEC_LAYERS), ()), (self.PHASE_ENC[0], self._bn_names(self.PHASE_ENC[0])),
                (self.PHASE_ENC[1], self._bn_names(self.PHASE_ENC[1]))]

    def phase_ranges(self):
        return [self.flat_range(self.DEC_LAYERS), self.flat_range(self.PHASE_ENC[0], self._bn_names(self.PHASE_ENC[0])),
                self.flat_range(self.PHASE_ENC[1], self._bn_names(self.PHASE_ENC[1]))]

    def backward_phases(self, gflows):
        """decoder | conv6_1..conv4 | conv3_1, conv_redir, cost volume, siamese conv3..conv1: in parameter order the tail,
        the middle and the head of the flat gradient buffer (same contract as FlowNetSEngine.backward_phases)."""
        self._ensure_grad_buffers()
        c, dc = self.cat, self.dcat
        g = list(gflows) + [None] * (5 - len(gflows))
        cb = self.chain_backward

        def decoder():
            self.decoder_backward({2: g[0], 3: g[1], 4: g[2], 5: g[3], 6: g[4]}, None)
            self.join_side()
            self.unpack_grads(self.DEC_LAYERS)

        def deep():
            cb("conv6_1", self.a6, self.a61, self.da6, False, self.da61)
            cb("conv6", c[5].slice(0, 512), self.a6, dc[5].slice(0, 512), True, self.da6)
            cb("conv5_1", self.a5, c[5].slice(0, 512), self.da5, False, dc[5].slice(0, 512))
            cb("conv5", c[4].slice(0, 512), self.a5, dc[4].slice(0, 512), True, self.da5)
            cb("conv4_1", self.a4, c[4].slice(0, 512), self.da4, False, dc[4].slice(0, 512))
            cb("conv4", c[3].slice(0, 256), self.a4, dc[3].slice(0, 256), True, self.da4)
            self.join_side()
            self.unpack_grads(self.PHASE_ENC[0])

        def shallow():
            cb("conv3_1", self.in31, c[3].slice(0, 256), self.din31, False, dc[3].slice(0, 256))
            # din31 = [d conv_redir out (32) | d corr out (441)]
            cb("conv_redir", self.c3["a"], self.in31.slice(0, 32), self.dc3["a"], False, self.din31.slice(0, 32))
            gcorr = self.din31.slice(32, 441)
            lrelu_bwd(gcorr, self.in31.slice(32, 441), SLOPE, self.ws)
            fa, fb = self.c3["a"], self.c3["b"]
            correlation_bwd_views(gcorr, fa, fb, self.dc3["a"], self.dc3["b"], 256, 256, 20, 2, 1, 0, self.ws.code)
            # siamese streams: same weights, wgrad slot per stream, shared BatchNorm parameter gradients accumulate
            cb("conv3", c[2].slice(0, 128), fa, dc[2].slice(0, 128), True, self.dc3["a"], "conv3@a", "conv3@a", 0, False)
            cb("conv2", self.c1["a"], c[2].slice(0, 128), self.dc1["a"], False, dc[2].slice(0, 128), "conv2@a", "conv2@a", 0, False)
            cb("conv1", self.xa, self.c1["a"], None, False, self.dc1["a"], "conv1@a", "conv1@a", 0, False)
            cb("conv3", self.c2b, fb, self.dc2b, False, self.dc3["b"], "conv3@b", "conv3@b", 1, True)
            cb("conv2", self.c1["b"], self.c2b, self.dc1["b"], False, self.dc2b, "conv2@b", "conv2@b", 1, True)
            cb("conv1", self.xb, self.c1["b"], None, False, self.dc1["b"], "conv1@b", "conv1@b", 1, True)
            self.join_side()
            self.unpack_grads(self.PHASE_ENC[1])
        return [decoder, deep, shallow]

    def backward(self, gflows) -> None:
        """gflows: gradients wrt (flow2, flow3, flow4, flow5, flow6) as (B,2,h,w) fp32 or None."""
        for phase in self.backward_phases(gflows):
            phase()


class _FlowNetCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        eng = module.engine_for(x)
        ctx.eng, ctx.module = eng, module
        out = tuple(eng.forward(x, module.training))
        if module.training:
            count_bn_batches(eng)
        return out

    @staticmethod
    def backward(ctx, *g):
        eng = ctx.eng
        eng.autograd_backward(g if eng.training_cache else (g[0], None, None, None, None), ctx.module._findex)
        table = eng.param_grads()
        grads = grads_for_autograd(ctx.module.parameters(), table)
        return (None, None) + grads


class FlowNetC(nn.Module, PackedOptimizerHook):
    """Drop-in for flownet2.networks.FlowNetC.FlowNetC(args, batchNorm=True, div_flow=20)."""

    def __init__(self, args=None, batchNorm: bool = True, div_flow: float = 20, precision: str = "bf16"):
        super().__init__()
        self.batchNorm, self.div_flow, self.precision = batchNorm, div_flow, precision
        self.conv1 = conv_block(batchNorm, 1, 64, 7, 2)
        self.conv2 = conv_block(batchNorm, 64, 128, 5, 2)
        self.conv3 = conv_block(batchNorm, 128, 256, 5, 2)
        self.conv_redir = conv_block(batchNorm, 256, 32, 1, 1)
        self.corr = Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2, corr_multiply=1)
        self.corr_activation = nn.LeakyReLU(SLOPE, inplace=True)
        self.conv3_1 = conv_block(batchNorm, 473, 256)
        for name, cin, cout, k, s in ENCODER[4:]:
            setattr(self, name, conv_block(batchNorm, cin, cout, k, s))
        for lvl, (cin, cout) in DECONV.items():
            setattr(self, f"deconv{lvl}", nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True),
                                                       nn.LeakyReLU(SLOPE, inplace=True)))
        for lvl, cin in PREDICT.items():
            setattr(self, f"predict_flow{lvl}", nn.Conv2d(cin, 2, 3, 1, 1, bias=True))
        for lvl in (6, 5, 4, 3):
            setattr(self, f"upsampled_flow{lvl}_to_{lvl - 1}", nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=True))
        for m in self.modules():  # flownet2/networks/FlowNetC.py:58-67
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if m.bias is not None:
                    nn.init.uniform_(m.bias)
                nn.init.xavier_uniform_(m.weight)
        self.upsample1 = nn.Upsample(scale_factor=4, mode="bilinear")
        self._engines: Dict[tuple, FlowNetCEngine] = {}

    def engine_for(self, x: torch.Tensor) -> FlowNetCEngine:
        if not x.is_cuda:
            raise RuntimeError("mireg.FlowNetC runs on the MI355X only; there is no CPU fallback")
        dtype = torch.bfloat16 if self.precision == "bf16" else torch.float32
        key = (tuple(x.shape), x.device, dtype, next(self.parameters()).data_ptr())
        if key not in self._engines:
            drop_engines(self)
            B, C, H, W = x.shape
            if C != 2:
                raise RuntimeError(f"FlowNetC expects (B,2,H,W) [fixed, moving], got {tuple(x.shape)}")
            self._engines[key] = FlowNetCEngine(self, B, H, W, x.device, dtype)
            install_bn_counter_hook(self)
        return self._engines[key]

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return tuple(_FlowNetCFn.apply(self, x.float(), *self.parameters()))
        eng = self.engine_for(x)
        out = tuple(eng.forward(x.float(), self.training))
        if self.training:
            count_bn_batches(eng)
        return out
