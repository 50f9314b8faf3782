"""Integer consumers of the quantizers' codes (extension; SURVEY §8(f) row 2 -- not part of the reference API).

The reference re-quantizes a wrapped layer's weight on every forward and then runs the float32 layer on the
dequantized tensors (``PytorchQuantizationWrapper.forward``, pytorch/quantize_wrapper.py:231-257, fed by
``PytorchActivationQuantizationHolder.forward``, pytorch/activation_quantization_holder.py:53).  Both operands of
that product are integers times a scale, so on MI355X the product itself can run on the 8-bit clamp indices:

    y[m][n] = float(sum_k (qa[m][k] - za) * qw[n][k]) * (sa * sw[n]) + bias[n]

``QuantizedLinear`` keeps the weight's int8 codes (refreshed when the weight tensor changes), turns the incoming
activation into codes with the activation quantizer's own parameters (``mctq_fq_codes_per_tensor``) and calls
``mctq_qlinear_i8`` (include/mctq_hip.h): 1 byte per weight streamed instead of 4 B read + 4 B written by the
fake-quant kernel and 4 B read again by the float32 GEMM.  The result is the exact integer sum scaled once; it
differs from the reference's float32 product only by that product's own accumulation rounding.

CPU tensors run the same integer arithmetic with torch ops (host logic for tests, bit-identical to the kernel).
"""
from typing import Optional

import torch
import torch.nn as nn

from mct_quantizers_amd.hip import native, ops
from mct_quantizers_amd.pytorch.containers import PytorchActivationQuantizationHolder, PytorchQuantizationWrapper

_MAX_K = 32768


def _activation_code_params(q):
    """(scale, zero_point, qmin, qmax) of an affine activation quantizer."""
    if hasattr(q, "scale") and hasattr(q, "zero_point"):              # ActivationUniform
        return float(q.scale), int(q.zero_point), q.min_quantized_domain, q.max_quantized_domain
    if hasattr(q, "scales") and hasattr(q, "zero_points") and not isinstance(q.scales, torch.Tensor):
        return float(q.scales), int(q.zero_points), q.min_quantized_domain, q.max_quantized_domain
    raise TypeError(f"{type(q).__name__} is not an affine per-tensor activation quantizer")


def _check_consumer_operands(a_codes, w_scales, w_rowsum, bias):
    """The kernels read w_scales / bias as float32 and w_rowsum as int32, all on a_codes' device."""
    for name, t, dt in (("w_scales", w_scales, torch.float32), ("w_rowsum", w_rowsum, torch.int32), ("bias", bias, torch.float32)):
        if t is None:
            continue
        if t.dtype != dt or t.device != a_codes.device or not t.is_contiguous():
            raise TypeError(f"{name} must be a contiguous {dt} tensor on {a_codes.device}, got {t.dtype} on {t.device}")


def qlinear_i8(a_codes: torch.Tensor, a_zero_point: int, a_scale: float, w_codes: torch.Tensor,
               w_scales: torch.Tensor, w_rowsum: torch.Tensor, bias: Optional[torch.Tensor],
               out_codes=None) -> torch.Tensor:
    """a_codes [M, K] int8/uint8, w_codes [N, K] int8 (zero point 0) -> float32 [M, N]; with
    ``out_codes = (scale, zero_point, qmin, qmax)`` the result leaves as the codes of that activation quantizer
    (int8 / uint8 [M, N]), bit-identical to quantizing the float32 result with ``ops.fq_codes``."""
    M, K = a_codes.shape
    N = w_codes.shape[0]
    if w_codes.shape[1] != K:
        raise RuntimeError(f"shape mismatch: activations have K={K}, weights K={w_codes.shape[1]}")
    if a_codes.is_cuda:
        if K % 16 or K > _MAX_K:
            raise NotImplementedError(f"mctq_qlinear_i8 needs K % 16 == 0 and K <= {_MAX_K}, got K={K}")
        _check_consumer_operands(a_codes, w_scales, w_rowsum, bias)
        lib = native.load()
        a_codes, w_codes = a_codes.contiguous(), w_codes.contiguous()
        code = native.CODE_U8 if a_codes.dtype == torch.uint8 else native.CODE_I8
        bias_ptr = None if bias is None else bias.data_ptr()
        with ops._maybe_on_device(a_codes):
            if out_codes is None:
                y = torch.empty((M, N), dtype=torch.float32, device=a_codes.device)
                rc = ops._launch(lib.mctq_qlinear_i8, a_codes.data_ptr(), code, int(a_zero_point), float(a_scale),
                                 w_codes.data_ptr(), w_scales.data_ptr(), w_rowsum.data_ptr(), bias_ptr, y.data_ptr(),
                                 M, N, K, ops._stream(a_codes))
            else:
                o_scale, o_zp, o_qmin, o_qmax = out_codes
                tdt, ocode = ops._code_dtype(o_qmin, o_qmax)
                y = torch.empty((M, N), dtype=tdt, device=a_codes.device)
                rc = ops._launch(lib.mctq_qlinear_i8_codes, a_codes.data_ptr(), code, int(a_zero_point), float(a_scale),
                                 w_codes.data_ptr(), w_scales.data_ptr(), w_rowsum.data_ptr(), bias_ptr, y.data_ptr(),
                                 ocode, float(o_scale), int(o_zp), int(o_qmin), int(o_qmax), M, N, K,
                                 ops._stream(a_codes))
        if rc:
            native.check(rc, "mctq_qlinear_i8")
        return y
    ops._cpu_route_allowed()
    acc = (a_codes.to(torch.int32) - int(a_zero_point)) @ w_codes.to(torch.int32).t()
    y = acc.to(torch.float32) * (torch.tensor(a_scale, dtype=torch.float64).to(torch.float32) * w_scales)
    if bias is not None:
        y = y + bias
    if out_codes is None:
        return y
    o_scale, o_zp, o_qmin, o_qmax = out_codes
    return ops.fq_codes(y, None, None, None, o_qmin, o_qmax, o_scale, o_zp)


def pack_w4(codes: torch.Tensor) -> torch.Tensor:
    """int8 codes in [-8, 7], [N, K] with K % 8 == 0 -> the consumer's 4-bit layout, uint8 [N, K / 2]: per group of 8
    consecutive k, byte j = (code[j] & 0xF) | (code[j + 4] << 4) (include/mctq_hip.h: mctq_qlinear_w4a8)."""
    N, K = codes.shape
    g = codes.reshape(N, K // 8, 8).to(torch.int16) & 0xF
    return (g[..., 0:4] | (g[..., 4:8] << 4)).to(torch.uint8).reshape(N, K // 2).contiguous()


def qlinear_w4a8(a_codes: torch.Tensor, a_zero_point: int, a_scale: float, w_codes4: torch.Tensor,
                 w_scales: torch.Tensor, w_rowsum: torch.Tensor, bias: Optional[torch.Tensor],
                 out_codes=None) -> torch.Tensor:
    """As ``qlinear_i8`` with the weights as packed 4-bit codes (``pack_w4``); GPU tensors only."""
    M, K = a_codes.shape
    N = w_codes4.shape[0]
    if w_codes4.shape[1] * 2 != K:
        raise RuntimeError(f"shape mismatch: activations have K={K}, packed weights K={w_codes4.shape[1] * 2}")
    if K % 16 or K > _MAX_K:
        raise NotImplementedError(f"mctq_qlinear_w4a8 needs K % 16 == 0 and K <= {_MAX_K}, got K={K}")
    _check_consumer_operands(a_codes, w_scales, w_rowsum, bias)
    lib = native.load()
    a_codes, w_codes4 = a_codes.contiguous(), w_codes4.contiguous()
    code = native.CODE_U8 if a_codes.dtype == torch.uint8 else native.CODE_I8
    if out_codes is None:
        y = torch.empty((M, N), dtype=torch.float32, device=a_codes.device)
        ocode, o_scale, o_zp, o_qmin, o_qmax = -1, 1.0, 0, 0, 0
    else:
        o_scale, o_zp, o_qmin, o_qmax = out_codes
        tdt, ocode = ops._code_dtype(o_qmin, o_qmax)
        y = torch.empty((M, N), dtype=tdt, device=a_codes.device)
    with ops._maybe_on_device(a_codes):
        rc = ops._launch(lib.mctq_qlinear_w4a8, a_codes.data_ptr(), code, int(a_zero_point), float(a_scale),
                         w_codes4.data_ptr(), w_scales.data_ptr(), w_rowsum.data_ptr(),
                         None if bias is None else bias.data_ptr(), y.data_ptr(), ocode, float(o_scale), int(o_zp),
                         int(o_qmin), int(o_qmax), M, N, K, ops._stream(a_codes))
    if rc:
        native.check(rc, "mctq_qlinear_w4a8")
    return y


_W4_MAX_ROWS = 32          # measured: beyond this the int8 kernels are faster than streaming half the bytes


class QuantizedLinear(nn.Module):
    """``activation quantizer -> PytorchQuantizationWrapper(nn.Linear)`` evaluated on integer codes.

    ``weights_quantizer``: WeightsSymmetric / WeightsPOT (zero point 0), per tensor or per output channel
    (``channel_axis`` 0), at most 8 bits.  ``activation_quantizer``: ActivationSymmetric / POT / Uniform, at most
    8 bits.  The float weight stays the module's parameter; its codes are rebuilt when it changes."""

    def __init__(self, linear: nn.Linear, weights_quantizer, activation_quantizer):
        super().__init__()
        if not isinstance(linear, nn.Linear):
            raise TypeError("QuantizedLinear wraps torch.nn.Linear")
        if not hasattr(weights_quantizer, "quantize_to_codes") or not hasattr(weights_quantizer, "threshold_np"):
            raise TypeError("the weights quantizer must be symmetric or power-of-two (zero point 0)")
        if weights_quantizer.per_channel and weights_quantizer.channel_axis % 2 != 0:
            raise NotImplementedError("per-channel weight scales must run along the output channels (axis 0)")
        if weights_quantizer.num_bits > 8 or activation_quantizer.num_bits > 8:
            raise NotImplementedError("codes wider than 8 bits")
        # The kernels read the bias as float32 and return float32: a half-precision layer stays on the fake-quant
        # path (its wrapper returns the layer's own type), it is never reinterpreted.
        if linear.weight.dtype != torch.float32 or (linear.bias is not None and linear.bias.dtype != torch.float32):
            raise TypeError(f"QuantizedLinear takes float32 layers, got weight {linear.weight.dtype}"
                            + ("" if linear.bias is None else f" / bias {linear.bias.dtype}"))
        if linear.bias is not None and linear.bias.device != linear.weight.device:
            raise TypeError("weight and bias live on different devices")
        self.weight = linear.weight
        self.bias = linear.bias
        self.in_features, self.out_features = linear.in_features, linear.out_features
        self.weights_quantizer = weights_quantizer
        self.activation_quantizer = activation_quantizer
        self._a_scale, self._a_zp, self._a_qmin, self._a_qmax = _activation_code_params(activation_quantizer)
        self._w_key = None
        self._w_codes = self._w_scales = self._w_rowsum = self._w_codes4 = None
        # chaining (fuse_linear_consumers(chain=True)): parameters of the activation quantizer that would quantize
        # this layer's output next; the output then leaves as that quantizer's codes
        self.emit_codes_for = None

    @classmethod
    def from_wrapper(cls, wrapper: PytorchQuantizationWrapper, activation_quantizer) -> "QuantizedLinear":
        quantizers = wrapper.weights_quantizers
        if list(quantizers) != ["weight"] or not isinstance(wrapper.layer, nn.Linear):
            raise TypeError("expected a wrapped torch.nn.Linear with one quantizer on 'weight'")
        lin = nn.Linear(wrapper.layer.in_features, wrapper.layer.out_features, bias=wrapper.layer.bias is not None,
                        device=wrapper.weight.device, dtype=wrapper.weight.dtype)
        lin.weight = wrapper.weight                     # the wrapper owns the float weight as its parameter
        lin.bias = wrapper.layer.bias
        return cls(lin, quantizers["weight"], activation_quantizer)

    def _refresh_weight_codes(self):
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        if key == self._w_key:
            return
        codes, scales, _ = self.weights_quantizer.quantize_to_codes(w.detach())
        if codes.dtype != torch.int8:
            raise RuntimeError("symmetric weight codes are expected to be int8")
        codes = codes.reshape(self.out_features, self.in_features)       # [O, C, 1, 1] of a pointwise convolution too
        scales = scales.to(device=w.device, dtype=torch.float32).reshape(-1)
        if scales.numel() == 1:
            scales = scales.expand(self.out_features)
        self._w_codes = codes.contiguous()
        self._w_scales = scales.contiguous()
        self._w_rowsum = codes.sum(dim=1, dtype=torch.int32).contiguous()
        # weights of at most 4 bits: also keep them packed, to stream half the bytes when there are few rows
        self._w_codes4 = pack_w4(self._w_codes) if (self.weights_quantizer.num_bits <= 4 and w.is_cuda
                                                     and self.in_features % 16 == 0) else None
        self._w_key = key

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._refresh_weight_codes()
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.in_features)
        if x2.dtype in (torch.uint8, torch.int8):          # already this layer's activation codes (chained layers)
            if x2.dtype != ops._code_dtype(self._a_qmin, self._a_qmax)[0]:
                raise TypeError(f"activation codes of type {x2.dtype} do not match this layer's quantizer")
            a_codes = x2
        else:
            a_codes = ops.fq_codes(x2, None, None, None, self._a_qmin, self._a_qmax, self._a_scale, self._a_zp)
        bias = None
        if self.bias is not None:
            bias = self.bias.detach()
            if bias.dtype != torch.float32 or bias.device != a_codes.device or not bias.is_contiguous():
                # (.to() after construction, e.g. model.half(): convert instead of letting the kernel misread it)
                bias = bias.to(device=a_codes.device, dtype=torch.float32).contiguous()
        if self._w_codes4 is not None and a_codes.is_cuda and a_codes.shape[0] <= _W4_MAX_ROWS:
            y = qlinear_w4a8(a_codes, self._a_zp, self._a_scale, self._w_codes4, self._w_scales, self._w_rowsum, bias,
                             self.emit_codes_for)
        else:
            y = qlinear_i8(a_codes, self._a_zp, self._a_scale, self._w_codes, self._w_scales, self._w_rowsum, bias,
                           self.emit_codes_for)
        return y.reshape(*lead, self.out_features)


class QuantizedConv1x1(QuantizedLinear):
    """``activation quantizer -> PytorchQuantizationWrapper(nn.Conv2d 1x1)`` on integer codes: a pointwise convolution
    (kernel 1x1, stride 1, no padding / dilation / groups -- most of the multiply-accumulates of MobileNet-style
    networks) is the same product over the channel axis for every pixel, so it runs on the same kernels with
    M = batch x height x width rows.  Channels-last inputs are quantized in place; NCHW inputs take one fused
    quantize-and-transpose pass (``mctq_fq_codes_nchw_to_nhwc``).  The result has the NCHW shape with channels-last strides."""

    def __init__(self, conv: nn.Conv2d, weights_quantizer, activation_quantizer):
        if not self.eligible(conv):
            raise TypeError("QuantizedConv1x1 takes 1x1, stride-1, unpadded, undilated, ungrouped nn.Conv2d layers")
        lin = nn.Linear(conv.in_channels, conv.out_channels, bias=conv.bias is not None, device=conv.weight.device,
                        dtype=conv.weight.dtype)
        lin.bias = conv.bias
        super().__init__(lin, weights_quantizer, activation_quantizer)
        self.weight = conv.weight                         # [O, C, 1, 1]; the codes are taken from it as [O, C]

    @staticmethod
    def eligible(conv) -> bool:
        return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
                and conv.padding in ((0, 0), 0, "valid") and conv.dilation == (1, 1) and conv.groups == 1
                and conv.padding_mode == "zeros")

    @classmethod
    def from_wrapper(cls, wrapper: PytorchQuantizationWrapper, activation_quantizer) -> "QuantizedConv1x1":
        quantizers = wrapper.weights_quantizers
        if list(quantizers) != ["weight"] or not cls.eligible(wrapper.layer):
            raise TypeError("expected a wrapped pointwise nn.Conv2d with one quantizer on 'weight'")
        conv = nn.Conv2d(wrapper.layer.in_channels, wrapper.layer.out_channels, 1, bias=wrapper.layer.bias is not None,
                         device=wrapper.weight.device, dtype=wrapper.weight.dtype)
        conv.weight = wrapper.weight
        conv.bias = wrapper.layer.bias
        return cls(conv, quantizers["weight"], activation_quantizer)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != self.in_features:
            raise RuntimeError(f"expected [N, {self.in_features}, H, W], got {tuple(x.shape)}")
        b, _, h, w_ = x.shape
        if x.dtype in (torch.uint8, torch.int8):          # codes from the previous layer: NCHW-shaped, NHWC-stored
            rows = x.permute(0, 2, 3, 1)
            rows = rows if rows.is_contiguous() else rows.contiguous()
        else:
            rows = ops.fq_codes_nhwc(x, self._a_qmin, self._a_qmax, self._a_scale, self._a_zp)
        y = super().forward(rows.reshape(b * h * w_, self.in_features))
        return y.reshape(b, h, w_, self.out_features).permute(0, 3, 1, 2)


def _consumer_for(wrapper, activation_quantizer):
    """The integer consumer that can stand in for ``wrapper`` fed by ``activation_quantizer``, or None."""
    layer = getattr(wrapper, "layer", None)
    if list(getattr(wrapper, "weights_quantizers", {})) != ["weight"]:
        return None
    weight = getattr(wrapper, "weight", None)
    bias = getattr(layer, "bias", None)
    if not isinstance(weight, torch.Tensor) or weight.dtype != torch.float32:
        return None                     # half-precision layers: the integer consumer would change the output type
    if isinstance(bias, torch.Tensor) and (bias.dtype != torch.float32 or bias.device != weight.device):
        return None
    try:
        if isinstance(layer, nn.Linear) and layer.in_features % 16 == 0 and layer.in_features <= _MAX_K:
            return QuantizedLinear.from_wrapper(wrapper, activation_quantizer)
        if QuantizedConv1x1.eligible(layer) and layer.in_channels % 16 == 0 and layer.in_channels <= _MAX_K:
            return QuantizedConv1x1.from_wrapper(wrapper, activation_quantizer)
    except (TypeError, NotImplementedError):
        return None
    return None


class _FusedAway(nn.Identity):
    """Placeholder left where an activation holder was folded into the QuantizedLinear after it."""


def _plain_holder(m) -> bool:
    """An activation holder that really quantizes (the FLN / preserving variants can be switched to pass-through)."""
    return isinstance(m, PytorchActivationQuantizationHolder) and not getattr(m, "quantization_bypass", False)


def fuse_linear_consumers(model: nn.Module, chain: bool = False) -> int:
    """In every ``nn.Sequential`` of ``model``: an activation holder directly followed by a wrapped ``nn.Linear`` with
    a symmetric weights quantizer becomes (Identity, QuantizedLinear).  Returns the number of pairs replaced.
    Pairs the integer consumer cannot take (other layers, LUT / uniform weights, K % 16 != 0) are left alone.

    ``chain=True``: where one QuantizedLinear feeds the next directly, the float32 tensor between them is never
    materialised -- the first emits the second's activation codes from its epilogue (same codes, bit for bit, as
    quantizing the float32 output).  Modules or hooks that look at that intermediate tensor then see uint8/int8 codes."""
    replaced = 0
    for seq in [m for m in model.modules() if isinstance(m, nn.Sequential)]:
        for i in range(len(seq) - 1):
            holder, wrapper = seq[i], seq[i + 1]
            if not _plain_holder(holder) or not isinstance(wrapper, PytorchQuantizationWrapper):
                continue
            fused = _consumer_for(wrapper, holder.activation_holder_quantizer)
            if fused is None:
                continue
            seq[i] = _FusedAway()
            seq[i + 1] = fused
            replaced += 1
        if chain:
            for i in range(len(seq) - 2):
                first, gap, second = seq[i], seq[i + 1], seq[i + 2]
                if isinstance(first, QuantizedLinear) and isinstance(gap, _FusedAway) and isinstance(second, QuantizedLinear):
                    first.emit_codes_for = (second._a_scale, second._a_zp, second._a_qmin, second._a_qmax)
    return replaced


def fuse_linear_consumers_fx(model: nn.Module, chain: bool = False):
    """The same rewrite on an arbitrary module graph (MCT-exported models are not ``nn.Sequential``): traces ``model``
    with torch.fx keeping wrappers and holders as leaves, and wherever an activation holder's ONLY consumer is a
    wrapped ``nn.Linear`` the integer consumer can take, replaces the pair by one ``QuantizedLinear`` node.
    Returns ``(graph_module, pairs_replaced)``.  Holders with several consumers (residual branches) stay."""
    import torch.fx as fx

    class _Tracer(fx.Tracer):
        def is_leaf_module(self, m, qualname):
            return isinstance(m, (PytorchQuantizationWrapper, PytorchActivationQuantizationHolder, QuantizedLinear)) \
                or super().is_leaf_module(m, qualname)

    graph = _Tracer().trace(model)
    gm = fx.GraphModule(model, graph)
    mods = dict(gm.named_modules())
    replaced = 0
    for node in list(gm.graph.nodes):
        if node.op != "call_module" or node.kwargs or len(node.args) != 1:
            continue
        wrapper = mods.get(node.target)
        if not isinstance(wrapper, PytorchQuantizationWrapper):
            continue
        src = node.args[0]
        if not isinstance(src, fx.Node) or src.op != "call_module" or len(src.users) != 1 or len(src.args) != 1 or src.kwargs:
            continue
        holder = mods.get(src.target)
        if not _plain_holder(holder):
            continue
        fused = _consumer_for(wrapper, holder.activation_holder_quantizer)
        if fused is None:
            continue
        name = node.target.replace(".", "_") + "_qlinear"
        gm.add_submodule(name, fused)
        mods[name] = fused
        node.target = name
        node.args = (src.args[0],)
        gm.graph.erase_node(src)
        replaced += 1
    if chain:
        for node in gm.graph.nodes:
            if node.op == "call_module" and isinstance(mods.get(node.target), QuantizedLinear) and len(node.users) == 1:
                user = next(iter(node.users))
                nxt = mods.get(user.target) if user.op == "call_module" else None
                if isinstance(nxt, QuantizedLinear) and user.args == (node,):
                    mods[node.target].emit_codes_for = (nxt._a_scale, nxt._a_zp, nxt._a_qmin, nxt._a_qmax)
    gm.graph.lint()
    gm.recompile()
    return gm, replaced
