"""``bench.py --config resnet50 --e2e``: one FORWARD of an MCT-export-shaped ResNet-50 per step.

The model (mct_quantizers_amd/workloads.py: wrapped_resnet50) is what MCT ships: 53 convolutions + the classifier under
``PytorchQuantizationWrapper`` (8-bit per-channel symmetric weights quantizers, weights re-quantized on EVERY forward,
reference pytorch/quantize_wrapper.py:228-240), an activation holder behind every ReLU (reference
pytorch/activation_quantization_holder.py:43-53).  It is saved with ``torch.save`` and loaded back with
``pytorch_load_quantized_model`` -- the reference's API, nothing else -- in three ways:

  per_layer      MCTQ_AUTO_BATCH=0: one quantizer launch per wrapped weight per forward, as the reference does
  auto_batched   the default: the loader installed ``accelerate`` -- ONE table launch for all 54 weights per forward
  auto_captured  MCTQ_AUTO_CAPTURE=1 at load time: ``model(x)`` itself replays its forward from a hipGraph (weights by one eager launch)
  captured       ``mq.accelerate(model, example_inputs=(x,))``: the whole forward replayed from one hipGraph

All three re-quantize the weights from their current float values on every forward.  Outputs are compared bit for bit
(same GPU, same convolution kernels); the quantized weights of the last forward are checked against the CPU oracle.
``value`` is forwards per second of the auto_batched mode; it is a side line in bench vocabulary, NOT the BASELINE
metric (that is ``bench.py`` without flags).
"""
from __future__ import annotations

import json
import os
import tempfile
import time

import torch

HBM_PEAK_GBS = 8000.0


def _timeit(fn, steps: int, warmup: int):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall * 1e3 / steps, e0.elapsed_time(e1) / steps


def main(args) -> int:
    assert torch.cuda.is_available(), "bench.py --e2e needs a GPU"
    torch.cuda.set_device(0)
    import mct_quantizers_amd as mq
    from mct_quantizers_amd import workloads
    from mct_quantizers_amd.hip import native
    native.load()
    assert native.fast() is not None, "the e2e bench needs the compiled binding"
    weights = "lut" if args.e2e_lut else "symmetric"
    batch, side = (args.batch if args.batch_given else 1), args.e2e_side
    steps, warmup = min(args.steps, 200), max(3, min(args.warmup, 20))
    x = torch.randn(batch, 3, side, side, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    path = os.path.join(tempfile.mkdtemp(prefix="mctq_e2e_"), "resnet50.pth")
    torch.save(workloads.wrapped_resnet50("cuda", weights=weights), path)

    def load(switch: str):
        old = os.environ.get("MCTQ_AUTO_BATCH")
        os.environ["MCTQ_AUTO_BATCH"] = switch
        try:
            return mq.pytorch_load_quantized_model(path)
        finally:
            if old is None:
                del os.environ["MCTQ_AUTO_BATCH"]
            else:
                os.environ["MCTQ_AUTO_BATCH"] = old

    def launches(model):
        n0 = native.launch_count()
        with torch.no_grad():
            y = model(x)
        return native.launch_count() - n0, y

    def quantized_weights(model):
        return [m.layer.weight.detach().clone() for m in model.modules() if isinstance(m, mq.PytorchQuantizationWrapper)]

    modes, outs, qws = {}, {}, {}
    with torch.no_grad():
        for name, switch in (("per_layer", "0"), ("auto_batched", "1")):
            model = load(switch)
            model(x)
            n, y = launches(model)
            wall_ms, dev_ms = _timeit(lambda: model(x), steps, warmup)
            modes[name] = {"ms_per_forward": wall_ms, "ms_per_forward_events": dev_ms, "quantizer_launches_per_forward": n}
            outs[name] = y.clone()
            qws[name] = quantized_weights(model)
            if name == "auto_batched":
                handle = mq.accelerated(model)
                assert handle is not None and handle._plan is not None
                # the weight launch alone (events on the launch stream): the roofline object
                q_ms = _timeit(handle.quantize_now, max(steps, 100), 10)[1]
                kernel = native.last_launch()
                w_model = model
    # MCTQ_AUTO_CAPTURE=1: the loaded model replays its own forward (graph per input signature, outputs cloned)
    os.environ["MCTQ_AUTO_CAPTURE"] = "1"
    try:
        auto_cap = load("1")
    finally:
        del os.environ["MCTQ_AUTO_CAPTURE"]
    with torch.no_grad():
        auto_cap(x); auto_cap(x)
        n_ac, y_ac = launches(auto_cap)
        wall_ms, dev_ms = _timeit(lambda: auto_cap(x), steps, warmup)
    modes["auto_captured"] = {"ms_per_forward": wall_ms, "ms_per_forward_events": dev_ms, "quantizer_launches_per_forward": n_ac,
                              "what": "MCTQ_AUTO_CAPTURE=1: model(x) runs its hooks eagerly (one weight launch) and replays the forward "
                                      "behind them from a hipGraph; outputs are clones"}
    outs["auto_captured"] = y_ac.clone()
    captured = mq.accelerate(load("0"), example_inputs=(x,))
    y_cap = captured(x).clone()
    wall_ms, dev_ms = _timeit(lambda: captured(x), steps, warmup)
    modes["captured"] = {"ms_per_forward": wall_ms, "ms_per_forward_events": dev_ms,
                         "what": "whole forward (one batched weight launch + layers + holders) replayed from one hipGraph"}
    outs["captured"] = y_cap

    n_weights = sum(int(xw.size) for xw, _ in workloads.make_model_weights("resnet50"))
    bytes_per_el = 8
    alg = n_weights * bytes_per_el
    q_us = q_ms * 1e3
    equal = {k: bool(torch.equal(outs["per_layer"], v)) for k, v in outs.items()}
    max_diff = {k: float((outs["per_layer"] - v).abs().max()) for k, v in outs.items()}
    rel_l2 = {k: float((outs["per_layer"] - v).norm() / outs["per_layer"].norm()) for k, v in outs.items()}
    weights_equal = len(qws["per_layer"]) == 54 and all(bool(torch.equal(a, b)) for a, b in zip(qws["per_layer"], qws["auto_batched"]))

    result = {
        "metric": f"forwards/s, MCT-export-shaped ResNet-50 ({'LUT' if weights == 'lut' else 'symmetric'} weights) loaded with "
                  f"pytorch_load_quantized_model, batch {batch} at {side}x{side}, all 54 wrapped weights re-quantized per forward",
        "value": 1e3 / modes["auto_batched"]["ms_per_forward"], "unit": "forwards/s", "n_gpus": 1, "ranks_seen": 1,
        "steps": steps, "warmup": warmup, "ms_per_step": modes["auto_batched"]["ms_per_forward"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (portable generator weights, torch.randn input; no checkpoint)",
        "config": {"workload": "e2e forward of wrapped ResNet-50 (NOT the BASELINE headline; see bench.py without flags)",
                   "weights_quantizer": "WeightsLUTSymmetricInferableQuantizer 16-entry" if weights == "lut"
                   else "WeightsSymmetricInferableQuantizer per-channel 8b",
                   "wrapped_weights": 54, "activation_holders": 49, "weight_elements": n_weights,
                   "batch": batch, "image": side, "launch": "eager forward; weights in one table launch (auto_batched)",
                   "loader": "mct_quantizers_amd.pytorch_load_quantized_model (reference pytorch/load_model.py:23-34)"},
        "modes": modes,
        "speedup_auto_batched_over_per_layer": modes["per_layer"]["ms_per_forward"] / modes["auto_batched"]["ms_per_forward"],
        "speedup_captured_over_per_layer": modes["per_layer"]["ms_per_forward"] / modes["captured"]["ms_per_forward"],
        "quantized_weights_bit_equal_per_layer_vs_auto_batched": weights_equal,
        "logits_bit_equal_to_per_layer": equal, "logits_max_abs_diff_to_per_layer": max_diff,
        "logits_relative_l2_diff_to_per_layer": rel_l2,
        "logits_note": "the quantizers' outputs are bit-equal between the modes; the convolutions around them (MIOpen) can answer "
                       "with another last bit when a bit-equal weight lives in another buffer (profiles/r04/conv_determinism_probe.log), "
                       "which later quantizers amplify to a flipped step -- logits are compared for information only",
        "roofline": {"bound": "hbm", "achieved": alg / q_us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / q_us / 1e3 / HBM_PEAK_GBS, "traffic": None, "kernel": kernel, "kernel_us": q_us,
                     "kernel_us_is": "average period of back-to-back weight launches of the loaded model's own handle "
                                     "(warm: one weight set, 102 MB, inside the Infinity Cache); the cold figure is "
                                     "bench.py --config resnet50",
                     "algorithmic_bytes_per_launch": alg},
    }
    if not args.no_cpu:
        # parity: the quantized weights the auto_batched model installed in its last forward == the CPU oracle's
        from oracle import torch_cpu
        stock = workloads.make_model_weights("resnet50")
        got = []
        for mod in w_model.modules():
            if isinstance(mod, mq.PytorchQuantizationWrapper):
                got.append(mod.layer.weight.detach().cpu())
        cls = "WeightsLUTSymmetricInferableQuantizer" if weights == "lut" else "WeightsSymmetricInferableQuantizer"
        same, c0, n = True, time.perf_counter(), 0
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        fs = []
        for (xw, kw), y in zip(stock, got):
            if weights == "lut":
                kw = dict(num_bits=4, lut_values=workloads.LUT16, threshold=kw["threshold"], per_channel=True,
                          channel_axis=0, input_rank=xw.ndim)
            f = torch_cpu.prepare(cls, kw)
            fs.append((f, torch.from_numpy(xw)))
            same = same and bool(torch.equal(f(fs[-1][1]), y))
        c0 = time.perf_counter()
        while time.perf_counter() - c0 < min(args.cpu_seconds, 10.0):
            for f, xc in fs:
                f(xc)
            n += 1
        el = time.perf_counter() - c0
        result["cpu_baseline"] = {"value": n_weights * n / el, "unit": "weight elems/s", "cores": torch.get_num_threads(),
                                  "kind": "port", "sample": f"{n} passes over the 54 weights with the CPU operator chain the "
                                                            f"reference runs (oracle/torch_cpu.py), {el:.1f} s",
                                  "gpu_output_bit_equal": same,
                                  "gpu_output_checked": "all 54 quantized weights of the auto_batched model's last forward"}
        if not same:
            result["parity_error"] = "quantized weights differ from the CPU oracle"
    if not weights_equal:
        result["parity_error"] = "quantized weights differ between the per-layer and the batched path"
    elif max(rel_l2.values()) > 0.05:
        result["parity_error"] = f"logits differ between modes by {max(rel_l2.values()):.3g} (relative L2)"
    print(json.dumps(result), flush=True)
    return 3 if result.get("parity_error") else 0
