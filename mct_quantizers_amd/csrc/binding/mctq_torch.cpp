// mctq_torch.cpp -- compiled Python binding of the hot entry points of libmctq_hip.so (C ABI: include/mctq_hip.h).
//
// Why: small activations are launch-bound -- per call the host cost IS the cost (BASELINE config 3).
// Through ctypes a call costs ~4.4 us before any tensor bookkeeping; this module does tensor checks, output
// allocation (torch's caching allocator), the stream lookup and the C-ABI call in one CPython FASTCALL.
// It adds no arithmetic and no kernels: every function ends in the same extern "C" entry point the ctypes
// binding (hip/native.py) calls, so the two bindings are interchangeable and the tests run both.
//
// Contract of every function: returns a new tensor, or Py_NotImplemented when the argument is not a plain,
// dense, supported-dtype HIP tensor in eager mode (CPU tensor, tensor subclass / FakeTensor, fx Proxy, an active
// torch.jit trace, gaps in the storage, float64 ...).  The Python caller (hip/ops.py) then takes the general route,
// which knows what the reference does in each of those cases.  A failing launch raises RuntimeError with
// mctq_last_error().
//
// Replaces, on the Python side of the boundary, the call sites
//   torch.fake_quantize_per_tensor_affine   activation_uniform_inferable_quantizer.py:124, activation_symmetric...py:113,
//                                           weights_symmetric_inferable_quantizer.py:147, weights_uniform...py:161
//   torch.fake_quantize_per_channel_affine  weights_symmetric_inferable_quantizer.py:139, weights_uniform...py:153
//   lut_quantizer                           pytorch/quantizer_utils.py:95-139
// (paths under /root/reference/mct_quantizers/pytorch/quantizers/).
//
// Host-only C++ (no device code); written against the HIP names of PyTorch-ROCm (c10::hip, at::detail::empty_*_cuda
// as exported by libtorch_hip).
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <ATen/core/Tensor.h>
#include <ATen/hip/EmptyTensor.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/Exceptions.h>
#include <torch/csrc/autograd/python_variable.h>
#include <torch/csrc/jit/frontend/tracer.h>
#include <torch/csrc/utils/python_arg_parser.h>

#include <vector>

#include "mctq_hip.h"

namespace {

inline int dtype_code(c10::ScalarType t) {
  switch (t) {
    case c10::ScalarType::Float: return MCTQ_DT_F32;
    case c10::ScalarType::Half: return MCTQ_DT_F16;
    case c10::ScalarType::BFloat16: return MCTQ_DT_BF16;
    case c10::ScalarType::Double: return MCTQ_DT_F64;
    default: return -1;
  }
}

// A plain eager HIP tensor the kernels can take as it is?  (Parameter is allowed: weights arrive as one.)
inline const at::Tensor* eligible(PyObject* obj, int* dt) {
  // exact types only (torch.Tensor, nn.Parameter): no subclass, no __torch_function__ -- and no isinstance() call through
  // the tensor metaclass on the hot path
  PyTypeObject* tp = Py_TYPE(obj);
  if (tp != (PyTypeObject*)THPVariableClass && tp != (PyTypeObject*)ParameterClass) return nullptr;
  const at::Tensor& x = THPVariable_Unpack(obj);
  if (!x.is_cuda() || x.layout() != c10::kStrided) return nullptr;
  *dt = dtype_code(x.scalar_type());
  if (*dt < 0) return nullptr;
  if (!x.unsafeGetTensorImpl()->is_non_overlapping_and_dense()) return nullptr;
  if (torch::jit::tracer::isTracing()) return nullptr;
  return &x;
}

// Output with the input's sizes AND strides: what ATen's empty_like (preserve format) gives for a non-overlapping
// dense tensor -- including the arbitrary strides of size-1 dimensions, which the reference's outputs keep.
inline at::Tensor like(const at::Tensor& x, c10::ScalarType dtype) {
  return at::Tensor(at::detail::empty_strided_cuda(x.sizes(), x.strides(), dtype, x.device()));
}

struct DeviceScope {          // make x's device current for the launch only when it is not already
  c10::DeviceIndex prev = -1;
  explicit DeviceScope(c10::DeviceIndex idx) {
    const c10::DeviceIndex cur = c10::hip::current_device();
    if (cur != idx) { prev = cur; c10::hip::set_device(idx); }
  }
  ~DeviceScope() { if (prev >= 0) c10::hip::set_device(prev); }
};

inline PyObject* raise_rc(int rc, const char* what) {
  PyErr_Format(PyExc_RuntimeError, "%s failed (rc=%d): %s", what, rc, mctq_last_error());
  return nullptr;
}

inline PyObject* not_implemented() { Py_RETURN_NOTIMPLEMENTED; }

// (outer, channels, inner) of a dense tensor in storage order for logical dimension `axis` (hip/ops.py:_channel_view)
inline void channel_view(const at::Tensor& x, int64_t axis, int64_t* outer, int64_t* c, int64_t* inner) {
  const int64_t n = x.numel();
  *c = x.size(axis);
  if (x.is_contiguous()) {
    int64_t in = 1;
    for (int64_t d = axis + 1; d < x.dim(); ++d) in *= x.size(d);
    *inner = in;
  } else {
    *inner = *c > 1 ? x.stride(axis) : 1;
  }
  const int64_t per = *c * *inner;
  *outer = per != 0 ? n / per : 0;
}

// A parameter vector usable as it is: float32 / int32, contiguous, on x's device, `want` elements.
inline const at::Tensor* param_tensor(PyObject* obj, const at::Tensor& x, c10::ScalarType dtype, int64_t want) {
  if (!THPVariable_Check(obj)) return nullptr;
  const at::Tensor& t = THPVariable_Unpack(obj);
  if (t.scalar_type() != dtype || !t.is_contiguous() || t.device() != x.device() || (want >= 0 && t.numel() != want))
    return nullptr;
  return &t;
}

inline bool as_double(PyObject* o, double* v) {
  *v = PyFloat_AsDouble(o);
  return !(*v == -1.0 && PyErr_Occurred());
}
inline bool as_i64(PyObject* o, int64_t* v) {
  *v = PyLong_AsLongLong(o);
  return !(*v == -1 && PyErr_Occurred());
}

// ---- per-tensor affine ---------------------------------------------------------------------------------
PyObject* launch_fq_per_tensor(PyObject* xo, float scale, int32_t zp, int32_t qmin, int32_t qmax) {
  HANDLE_TH_ERRORS
  int dt;
  const at::Tensor* xp = eligible(xo, &dt);
  if (!xp) return not_implemented();
  const at::Tensor& x = *xp;
  if (qmin > qmax) {                      // ATen's checks and messages (fake_quantize_per_tensor_affine), in its order
    PyErr_SetString(PyExc_RuntimeError, "`quant_min` should be less than or         equal to `quant_max`.");
    return nullptr;
  }
  if (zp < qmin || zp > qmax) {
    PyErr_SetString(PyExc_RuntimeError, "`zero_point` must be between `quant_min` and `quant_max`.");
    return nullptr;
  }
  at::Tensor y = like(x, x.scalar_type());
  const c10::DeviceIndex idx = x.device().index();
  DeviceScope scope(idx);
  const int rc = mctq_fq_per_tensor(x.const_data_ptr(), y.mutable_data_ptr(), x.numel(), dt, scale, zp, qmin, qmax,
                                    (void*)c10::hip::getCurrentHIPStream(idx).stream());
  if (rc) return raise_rc(rc, "mctq_fq_per_tensor");
  return THPVariable_Wrap(std::move(y));
  END_HANDLE_TH_ERRORS
}

// fq_per_tensor(x, scale, zero_point, quant_min, quant_max)
PyObject* py_fq_per_tensor(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 5) { PyErr_SetString(PyExc_TypeError, "fq_per_tensor(x, scale, zero_point, quant_min, quant_max)"); return nullptr; }
  double scale; int64_t zp, qmin, qmax;
  if (!as_double(args[1], &scale) || !as_i64(args[2], &zp) || !as_i64(args[3], &qmin) || !as_i64(args[4], &qmax)) return nullptr;
  return launch_fq_per_tensor(args[0], (float)scale, (int32_t)zp, (int32_t)qmin, (int32_t)qmax);
}

// ---- per-channel affine --------------------------------------------------------------------------------
PyObject* launch_fq_per_channel(PyObject* xo, PyObject* scales_o, PyObject* zps_o, int64_t axis, int32_t qmin, int32_t qmax) {
  HANDLE_TH_ERRORS
  int dt;
  const at::Tensor* xp = eligible(xo, &dt);
  if (!xp) return not_implemented();
  const at::Tensor& x = *xp;
  if (axis < 0 || axis >= x.dim()) return not_implemented();            // the general route raises ATen's message
  const at::Tensor* sp = param_tensor(scales_o, x, c10::ScalarType::Float, x.size(axis));
  if (!sp) return not_implemented();
  const at::Tensor* zp = nullptr;
  if (zps_o != Py_None) {
    zp = param_tensor(zps_o, x, c10::ScalarType::Int, x.size(axis));
    if (!zp) return not_implemented();
  }
  at::Tensor y = like(x, x.scalar_type());
  int64_t outer, c, inner;
  channel_view(x, axis, &outer, &c, &inner);
  const c10::DeviceIndex idx = x.device().index();
  DeviceScope scope(idx);
  const int rc = mctq_fq_per_channel(x.const_data_ptr(), y.mutable_data_ptr(), outer, c, inner, dt, sp->const_data_ptr<float>(),
                                     zp ? zp->const_data_ptr<int32_t>() : nullptr, qmin, qmax,
                                     (void*)c10::hip::getCurrentHIPStream(idx).stream());
  if (rc) return raise_rc(rc, "mctq_fq_per_channel");
  return THPVariable_Wrap(std::move(y));
  END_HANDLE_TH_ERRORS
}

// fq_per_channel(x, scales, zero_points | None, axis, quant_min, quant_max)
PyObject* py_fq_per_channel(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 6) { PyErr_SetString(PyExc_TypeError, "fq_per_channel(x, scales, zero_points, axis, quant_min, quant_max)"); return nullptr; }
  int64_t axis, qmin, qmax;
  if (!as_i64(args[3], &axis) || !as_i64(args[4], &qmin) || !as_i64(args[5], &qmax)) return nullptr;
  return launch_fq_per_channel(args[0], args[1], args[2], axis, (int32_t)qmin, (int32_t)qmax);
}

// ---- per-tensor affine, parameters read on the device ---------------------------------------------------
// fq_per_tensor_tqp(x, scale_tensor, zero_point_tensor, quant_min, quant_max): the tensor-qparams overload
// (weights_symmetric_inferable_quantizer.py:147-151 passes 1-element tensors); no device->host read.
PyObject* py_fq_per_tensor_tqp(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
  HANDLE_TH_ERRORS
  if (nargs != 5) { PyErr_SetString(PyExc_TypeError, "fq_per_tensor_tqp(x, scale, zero_point, quant_min, quant_max)"); return nullptr; }
  int64_t qmin, qmax;
  if (!as_i64(args[3], &qmin) || !as_i64(args[4], &qmax)) return nullptr;
  int dt;
  const at::Tensor* xp = eligible(args[0], &dt);
  if (!xp) return not_implemented();
  const at::Tensor& x = *xp;
  const at::Tensor* sp = param_tensor(args[1], x, c10::ScalarType::Float, 1);
  const at::Tensor* zp = param_tensor(args[2], x, c10::ScalarType::Int, 1);
  if (!sp || !zp) return not_implemented();
  at::Tensor y = like(x, x.scalar_type());
  const c10::DeviceIndex idx = x.device().index();
  DeviceScope scope(idx);
  const int rc = mctq_fq_per_tensor_tqp(x.const_data_ptr(), y.mutable_data_ptr(), x.numel(), dt, sp->const_data_ptr<float>(),
                                        zp->const_data_ptr<int32_t>(), (int32_t)qmin, (int32_t)qmax,
                                        (void*)c10::hip::getCurrentHIPStream(idx).stream());
  if (rc) return raise_rc(rc, "mctq_fq_per_tensor_tqp");
  return THPVariable_Wrap(std::move(y));
  END_HANDLE_TH_ERRORS
}

// ---- LUT, decision table ---------------------------------------------------------------------------------
// lutt_per_tensor(x, table, step_round, thr_div, thr_mul, mult, clip_min, clip_max) -> float32 tensor
PyObject* py_lutt_per_tensor(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
  HANDLE_TH_ERRORS
  if (nargs != 8) { PyErr_SetString(PyExc_TypeError, "lutt_per_tensor(x, table, step_round, thr_div, thr_mul, mult, clip_min, clip_max)"); return nullptr; }
  int64_t step_round; double thr_div, thr_mul, mult, cmin, cmax;
  if (!as_i64(args[2], &step_round) || !as_double(args[3], &thr_div) || !as_double(args[4], &thr_mul) ||
      !as_double(args[5], &mult) || !as_double(args[6], &cmin) || !as_double(args[7], &cmax)) return nullptr;
  int dt;
  const at::Tensor* xp = eligible(args[0], &dt);
  if (!xp || dt == MCTQ_DT_F64) return not_implemented();
  const at::Tensor& x = *xp;
  const at::Tensor* tp = param_tensor(args[1], x, c10::ScalarType::Float, -1);
  if (!tp || tp->dim() != 2 || tp->size(1) != 2) return not_implemented();
  at::Tensor y = like(x, c10::ScalarType::Float);
  const c10::DeviceIndex idx = x.device().index();
  DeviceScope scope(idx);
  const int rc = mctq_lutt_per_tensor(x.const_data_ptr(), y.mutable_data_ptr<float>(), x.numel(), dt, (int32_t)step_round,
      (float)thr_div, (float)thr_mul, tp->const_data_ptr<float>(), (int32_t)tp->size(0) - 1,
      (float)mult, (float)cmin, (float)cmax, (void*)c10::hip::getCurrentHIPStream(idx).stream());
  if (rc) return raise_rc(rc, "mctq_lutt_per_tensor");
  return THPVariable_Wrap(std::move(y));
  END_HANDLE_TH_ERRORS
}

// lutt_per_channel(x, thresholds, eps, table, axis, mult, clip_min, clip_max) -> float32 tensor
PyObject* py_lutt_per_channel(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
  HANDLE_TH_ERRORS
  if (nargs != 8) { PyErr_SetString(PyExc_TypeError, "lutt_per_channel(x, thresholds, eps, table, axis, mult, clip_min, clip_max)"); return nullptr; }
  int64_t axis; double eps, mult, cmin, cmax;
  if (!as_double(args[2], &eps) || !as_i64(args[4], &axis) || !as_double(args[5], &mult) || !as_double(args[6], &cmin) ||
      !as_double(args[7], &cmax)) return nullptr;
  int dt;
  const at::Tensor* xp = eligible(args[0], &dt);
  if (!xp || dt == MCTQ_DT_F64) return not_implemented();
  const at::Tensor& x = *xp;
  if (axis < 0 || axis >= x.dim()) return not_implemented();
  const at::Tensor* thr = param_tensor(args[1], x, c10::ScalarType::Float, x.size(axis));
  const at::Tensor* tp = param_tensor(args[3], x, c10::ScalarType::Float, -1);
  if (!thr || !tp || tp->dim() != 2 || tp->size(1) != 2) return not_implemented();
  at::Tensor y = like(x, c10::ScalarType::Float);
  int64_t outer, c, inner;
  channel_view(x, axis, &outer, &c, &inner);
  const c10::DeviceIndex idx = x.device().index();
  DeviceScope scope(idx);
  const int rc = mctq_lutt_per_channel(x.const_data_ptr(), y.mutable_data_ptr<float>(), outer, c, inner, dt,
      thr->const_data_ptr<float>(), (float)eps, tp->const_data_ptr<float>(),
      (int32_t)tp->size(0) - 1, (float)mult, (float)cmin, (float)cmax,
      (void*)c10::hip::getCurrentHIPStream(idx).stream());
  if (rc) return raise_rc(rc, "mctq_lutt_per_channel");
  return THPVariable_Wrap(std::move(y));
  END_HANDLE_TH_ERRORS
}

// ---- a list of weights in ONE launch ------------------------------------------------------------------
// fq_batched(items) with items = sequence of (x, scales, zero_points | None, axis | None, quant_min, quant_max);
// axis None = per tensor (scales / zero_points are 1-element device tensors).  Returns a list of new tensors,
// or NotImplemented if any item is not eligible (the caller then quantizes them one by one).
PyObject* py_fq_batched(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
  HANDLE_TH_ERRORS
  if (nargs != 1) { PyErr_SetString(PyExc_TypeError, "fq_batched(items)"); return nullptr; }
  PyObject* seq = PySequence_Fast(args[0], "fq_batched expects a sequence of tuples");
  if (!seq) return nullptr;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  std::vector<mctq_fq_item> items((size_t)n);
  std::vector<at::Tensor> outs;
  outs.reserve((size_t)n);
  c10::DeviceIndex idx = -1;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject* it = PySequence_Fast_GET_ITEM(seq, i);
    if (!PyTuple_Check(it) || PyTuple_GET_SIZE(it) != 6) {
      Py_DECREF(seq);
      PyErr_SetString(PyExc_TypeError, "fq_batched item: (x, scales, zero_points, axis, quant_min, quant_max)");
      return nullptr;
    }
    int dt;
    const at::Tensor* xp = eligible(PyTuple_GET_ITEM(it, 0), &dt);
    int64_t qmin, qmax, axis = -1;
    PyObject* axis_o = PyTuple_GET_ITEM(it, 3);
    if (!as_i64(PyTuple_GET_ITEM(it, 4), &qmin) || !as_i64(PyTuple_GET_ITEM(it, 5), &qmax) ||
        (axis_o != Py_None && !as_i64(axis_o, &axis))) { Py_DECREF(seq); return nullptr; }
    bool ok = xp != nullptr && (axis_o == Py_None || (axis >= 0 && axis < xp->dim()));
    const at::Tensor *sp = nullptr, *zp = nullptr;
    if (ok) {
      if (idx < 0) idx = xp->device().index();
      ok = xp->device().index() == idx;
      const int64_t want = axis_o == Py_None ? 1 : xp->size(axis);
      sp = param_tensor(PyTuple_GET_ITEM(it, 1), *xp, c10::ScalarType::Float, want);
      ok = ok && sp != nullptr;
      if (PyTuple_GET_ITEM(it, 2) != Py_None) {
        zp = param_tensor(PyTuple_GET_ITEM(it, 2), *xp, c10::ScalarType::Int, want);
        ok = ok && zp != nullptr;
      }
    }
    if (!ok) { Py_DECREF(seq); return not_implemented(); }
    outs.push_back(like(*xp, xp->scalar_type()));
    mctq_fq_item& d = items[(size_t)i];
    d.x = xp->const_data_ptr(); d.y = outs.back().mutable_data_ptr();
    if (axis_o == Py_None) { d.outer = 1; d.channels = 1; d.inner = xp->numel(); }
    else channel_view(*xp, axis, &d.outer, &d.channels, &d.inner);
    d.scales = sp->const_data_ptr<float>();
    d.zero_points = zp ? zp->const_data_ptr<int32_t>() : nullptr;
    d.quant_min = (int32_t)qmin; d.quant_max = (int32_t)qmax; d.dtype = dt;
    d.flags = axis_o == Py_None ? MCTQ_FQ_ITEM_PER_TENSOR : 0;
    if (dt == MCTQ_DT_F64 && axis_o == Py_None && !zp) { Py_DECREF(seq); return not_implemented(); }   // general route supplies zeros
  }
  Py_DECREF(seq);
  if (n > 0) {
    DeviceScope scope(idx);
    const int rc = mctq_fq_batched(items.data(), (int32_t)n, (void*)c10::hip::getCurrentHIPStream(idx).stream());
    if (rc) return raise_rc(rc, "mctq_fq_batched");
  }
  PyObject* list = PyList_New(n);
  if (!list) return nullptr;
  for (Py_ssize_t i = 0; i < n; ++i) PyList_SET_ITEM(list, i, THPVariable_Wrap(std::move(outs[(size_t)i])));
  return list;
  END_HANDLE_TH_ERRORS
}

// ---- pre-packed launch arguments: plan = AffinePlan(scale, zp, qmin, qmax) / AffinePlan(scales, zps|None, axis, qmin, qmax);
//      plan(x) -> tensor | NotImplemented -------------------------------------------------------------------
struct AffinePlan {
  PyObject_HEAD
  vectorcallfunc vectorcall;
  PyObject* scales;       // NULL: per tensor
  PyObject* zps;          // Py_None or tensor (per channel)
  int64_t axis;
  float scale;
  int32_t zp, qmin, qmax;
};

PyObject* plan_vectorcall(PyObject* self, PyObject* const* args, size_t nargsf, PyObject* kwnames) {
  AffinePlan* p = (AffinePlan*)self;
  if (PyVectorcall_NARGS(nargsf) != 1 || (kwnames && PyTuple_GET_SIZE(kwnames))) {
    PyErr_SetString(PyExc_TypeError, "AffinePlan.__call__(x)");
    return nullptr;
  }
  if (!p->scales) return launch_fq_per_tensor(args[0], p->scale, p->zp, p->qmin, p->qmax);
  return launch_fq_per_channel(args[0], p->scales, p->zps, p->axis, p->qmin, p->qmax);
}

PyObject* plan_new(PyTypeObject* type, PyObject* args, PyObject*) {
  const Py_ssize_t n = PyTuple_GET_SIZE(args);
  AffinePlan* p = (AffinePlan*)type->tp_alloc(type, 0);
  if (!p) return nullptr;
  p->vectorcall = plan_vectorcall;
  p->scales = nullptr; p->zps = nullptr; p->axis = 0;
  int64_t zp = 0, qmin, qmax;
  if (n == 4) {
    double scale;
    if (!as_double(PyTuple_GET_ITEM(args, 0), &scale) || !as_i64(PyTuple_GET_ITEM(args, 1), &zp) ||
        !as_i64(PyTuple_GET_ITEM(args, 2), &qmin) || !as_i64(PyTuple_GET_ITEM(args, 3), &qmax)) { Py_DECREF(p); return nullptr; }
    p->scale = (float)scale;
  } else if (n == 5) {
    if (!as_i64(PyTuple_GET_ITEM(args, 2), &p->axis) || !as_i64(PyTuple_GET_ITEM(args, 3), &qmin) ||
        !as_i64(PyTuple_GET_ITEM(args, 4), &qmax)) { Py_DECREF(p); return nullptr; }
    p->scales = PyTuple_GET_ITEM(args, 0); Py_INCREF(p->scales);
    p->zps = PyTuple_GET_ITEM(args, 1); Py_INCREF(p->zps);
  } else {
    Py_DECREF(p);
    PyErr_SetString(PyExc_TypeError, "AffinePlan(scale, zero_point, qmin, qmax) or AffinePlan(scales, zero_points, axis, qmin, qmax)");
    return nullptr;
  }
  p->zp = (int32_t)zp; p->qmin = (int32_t)qmin; p->qmax = (int32_t)qmax;
  return (PyObject*)p;
}

void plan_dealloc(PyObject* self) {
  AffinePlan* p = (AffinePlan*)self;
  Py_XDECREF(p->scales);
  Py_XDECREF(p->zps);
  Py_TYPE(self)->tp_free(self);
}

PyTypeObject AffinePlanType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// ---- the whole eager call of an activation holder in ONE C call ---------------------------------------------
// PytorchActivationQuantizationHolder.forward (reference: pytorch/activation_quantization_holder.py:43-53) is
// `return self.activation_holder_quantizer(inputs)`; nn.Module.__call__ around it serves hooks, compiled calls and
// tracing.  fast = HolderCall(holder.__dict__, quantizer, quantizer.__dict__, plan, (hook dicts ...), check_bypass)
// fast(x) -> tensor | NotImplemented: checks, in C, that none of that machinery is in use and that the objects the
// launch state was derived from are still the ones in place --
//   every dict in `hook dicts` is empty (the module's and torch's global forward / backward hook tables),
//   holder.__dict__ has no "_compiled_call_impl" other than None, holder.__dict__["activation_holder_quantizer"] is `quantizer`,
//   (check_bypass) holder.__dict__["quantization_bypass"] is False,
//   quantizer.__dict__["_plan"] is `plan` (assigning to a public parameter of the quantizer drops its plan) --
// and then makes the plan's call (which itself declines anything but a plain eager HIP tensor).
struct HolderCall {
  PyObject_HEAD
  vectorcallfunc vectorcall;
  PyObject* holder_dict;
  PyObject* quantizer;
  PyObject* quantizer_dict;
  PyObject* plan;
  PyObject* hook_dicts;     // tuple
  int check_bypass;
};

PyObject *kw_compiled = nullptr, *kw_quantizer = nullptr, *kw_bypass = nullptr, *kw_plan = nullptr;

PyObject* holdercall_vectorcall(PyObject* self, PyObject* const* args, size_t nargsf, PyObject* kwnames) {
  HolderCall* h = (HolderCall*)self;
  if (PyVectorcall_NARGS(nargsf) != 1 || (kwnames && PyTuple_GET_SIZE(kwnames))) {
    PyErr_SetString(PyExc_TypeError, "HolderCall.__call__(x)");
    return nullptr;
  }
  const Py_ssize_t nd = PyTuple_GET_SIZE(h->hook_dicts);
  for (Py_ssize_t i = 0; i < nd; ++i)
    if (PyDict_GET_SIZE(PyTuple_GET_ITEM(h->hook_dicts, i)) != 0) return not_implemented();
  // (a class attribute of nn.Module, None, until Module.compile() sets it on the instance)
  PyObject* compiled = PyDict_GetItem(h->holder_dict, kw_compiled);
  if (compiled != nullptr && compiled != Py_None) return not_implemented();
  if (PyDict_GetItem(h->holder_dict, kw_quantizer) != h->quantizer) return not_implemented();
  if (h->check_bypass && PyDict_GetItem(h->holder_dict, kw_bypass) != Py_False) return not_implemented();
  if (PyDict_GetItem(h->quantizer_dict, kw_plan) != h->plan) return not_implemented();
  return plan_vectorcall(h->plan, args, 1, nullptr);
}

PyObject* holdercall_new(PyTypeObject* type, PyObject* args, PyObject*) {
  PyObject *hd, *q, *qd, *plan, *hooks;
  int bypass = 0;
  if (!PyArg_ParseTuple(args, "O!OO!O!O!|p", &PyDict_Type, &hd, &q, &PyDict_Type, &qd, &AffinePlanType, &plan,
                        &PyTuple_Type, &hooks, &bypass)) return nullptr;
  for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(hooks); ++i)
    if (!PyDict_Check(PyTuple_GET_ITEM(hooks, i))) {
      PyErr_SetString(PyExc_TypeError, "HolderCall: the hook tables must be dicts");
      return nullptr;
    }
  HolderCall* h = (HolderCall*)type->tp_alloc(type, 0);
  if (!h) return nullptr;
  h->vectorcall = holdercall_vectorcall;
  h->holder_dict = hd; h->quantizer = q; h->quantizer_dict = qd; h->plan = plan; h->hook_dicts = hooks;
  h->check_bypass = bypass;
  Py_INCREF(hd); Py_INCREF(q); Py_INCREF(qd); Py_INCREF(plan); Py_INCREF(hooks);
  return (PyObject*)h;
}

int holdercall_traverse(PyObject* self, visitproc visit, void* arg) {
  HolderCall* h = (HolderCall*)self;
  Py_VISIT(h->holder_dict); Py_VISIT(h->quantizer); Py_VISIT(h->quantizer_dict); Py_VISIT(h->plan); Py_VISIT(h->hook_dicts);
  return 0;
}

int holdercall_clear(PyObject* self) {
  HolderCall* h = (HolderCall*)self;
  Py_CLEAR(h->holder_dict); Py_CLEAR(h->quantizer); Py_CLEAR(h->quantizer_dict); Py_CLEAR(h->plan); Py_CLEAR(h->hook_dicts);
  return 0;
}

void holdercall_dealloc(PyObject* self) {
  PyObject_GC_UnTrack(self);
  holdercall_clear(self);
  Py_TYPE(self)->tp_free(self);
}

PyTypeObject HolderCallType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// ---- pre-packed arguments of the per-tensor LUT quantizer with a decision table (the activation LUT quantizer):
//      plan = LutPlan(table, thr_div_f32, thr_div_f16, thr_div_bf16, thr_mul, mult, clip_min, clip_max, half_steps);
//      plan(x) -> float32 tensor | NotImplemented.  The divisor depends on the tensor's type because the reference
//      narrows the Python-float `threshold + eps` to it (activation_lut_pot_inferable_quantizer.py:86-91);
//      half_steps != 0: a half-precision tensor also rounds the quotient and the scaled value to its own type;
//      half_steps == 2: half-precision tensors are declined (NotImplemented) -- their clip bounds differ from the table's.
struct LutPlan {
  PyObject_HEAD
  vectorcallfunc vectorcall;
  PyObject* table;
  float thr_div[3];
  float thr_mul, mult, cmin, cmax;
  int half_steps;
};

PyObject* lutplan_vectorcall(PyObject* self, PyObject* const* args, size_t nargsf, PyObject* kwnames) {
  HANDLE_TH_ERRORS
  LutPlan* p = (LutPlan*)self;
  if (PyVectorcall_NARGS(nargsf) != 1 || (kwnames && PyTuple_GET_SIZE(kwnames))) {
    PyErr_SetString(PyExc_TypeError, "LutPlan.__call__(x)");
    return nullptr;
  }
  int dt;
  const at::Tensor* xp = eligible(args[0], &dt);
  if (!xp || dt == MCTQ_DT_F64) return not_implemented();
  if (p->half_steps == 2 && dt != MCTQ_DT_F32) return not_implemented();   // clip bounds not exact in a half type: general route
  const at::Tensor& x = *xp;
  const at::Tensor* tp = param_tensor(p->table, x, c10::ScalarType::Float, -1);
  if (!tp || tp->dim() != 2 || tp->size(1) != 2) return not_implemented();
  at::Tensor y = like(x, c10::ScalarType::Float);
  const c10::DeviceIndex idx = x.device().index();
  DeviceScope scope(idx);
  const int step = (p->half_steps && dt != MCTQ_DT_F32) ? dt : 0;
  const int rc = mctq_lutt_per_tensor(x.const_data_ptr(), y.mutable_data_ptr<float>(), x.numel(), dt, step, p->thr_div[dt],
      p->thr_mul, tp->const_data_ptr<float>(), (int32_t)tp->size(0) - 1, p->mult, p->cmin,
      p->cmax, (void*)c10::hip::getCurrentHIPStream(idx).stream());
  if (rc) return raise_rc(rc, "mctq_lutt_per_tensor");
  return THPVariable_Wrap(std::move(y));
  END_HANDLE_TH_ERRORS
}

PyObject* lutplan_new(PyTypeObject* type, PyObject* args, PyObject*) {
  if (PyTuple_GET_SIZE(args) != 9) {
    PyErr_SetString(PyExc_TypeError, "LutPlan(table, thr_div_f32, thr_div_f16, thr_div_bf16, thr_mul, mult, clip_min, clip_max, half_steps)");
    return nullptr;
  }
  LutPlan* p = (LutPlan*)type->tp_alloc(type, 0);
  if (!p) return nullptr;
  p->vectorcall = lutplan_vectorcall;
  p->table = nullptr;
  double v[7];
  for (int i = 0; i < 7; ++i)
    if (!as_double(PyTuple_GET_ITEM(args, 1 + i), &v[i])) { Py_DECREF(p); return nullptr; }
  int64_t hs;
  if (!as_i64(PyTuple_GET_ITEM(args, 8), &hs)) { Py_DECREF(p); return nullptr; }
  if (!THPVariable_Check(PyTuple_GET_ITEM(args, 0))) {
    Py_DECREF(p);
    PyErr_SetString(PyExc_TypeError, "LutPlan: table must be a tensor");
    return nullptr;
  }
  p->table = PyTuple_GET_ITEM(args, 0); Py_INCREF(p->table);
  p->thr_div[MCTQ_DT_F32] = (float)v[0]; p->thr_div[MCTQ_DT_F16] = (float)v[1]; p->thr_div[MCTQ_DT_BF16] = (float)v[2];
  p->thr_mul = (float)v[3]; p->mult = (float)v[4]; p->cmin = (float)v[5]; p->cmax = (float)v[6];
  p->half_steps = (int)hs;
  return (PyObject*)p;
}

void lutplan_dealloc(PyObject* self) {
  Py_XDECREF(((LutPlan*)self)->table);
  Py_TYPE(self)->tp_free(self);
}

PyTypeObject LutPlanType = {PyVarObject_HEAD_INIT(nullptr, 0)};

// ---- a whole list of weights, pre-packed: plan = BatchPlan(items) with items = sequence of
//      (x, y, scales, zero_points | None, axis | None, quant_min, quant_max[, watch]); plan() re-quantizes every x
//      into ITS y (caller-owned, persistent output buffers) with ONE launch per storage type, whatever the number of
//      tensors (mctq_fq_batch_pack / mctq_fq_batch_run: the descriptors live in a device table owned by the plan), and
//      returns None -- no allocation, no per-tensor Python.
//      Every call re-reads the tensors: the device pointers of x, y, scales and zero_points are followed (a Parameter
//      whose storage was swapped; the table is re-packed and re-uploaded then -- one small synchronous copy, not legal
//      under stream capture); a change of sizes, strides, dtype or device returns NotImplemented (the caller rebuilds
//      the plan), as does a stale `watch`.
//      watch = (dict, ((name, object, version), ...)): dict[name] must still BE `object` and, for version >= 0, that
//      tensor's in-place version counter must still equal `version` -- how the plan notices that a quantizer's public
//      parameters were replaced or edited in place since it was built.
//      Every launch bumps the version counter of every output tensor (the kernel rewrites it in place behind autograd's
//      back otherwise: a quantized weight saved for a backward and overwritten by a later forward must raise "modified by
//      an inplace operation", as any in-place op would).
//      BatchPlan(items, True) = VERSIONED REUSE ("quantize once, not per forward"; the reference's
//      enable_reuse_quantizer idea, base_pytorch_inferable_quantizer.py:34-49, made safe): a call whose inputs are what
//      the last launch read -- same device pointer, same in-place version counter, same dtype / sizes of every x, outputs
//      not written by anybody else since, parameters unchanged (watch) -- issues NO launch and returns None; any change
//      relaunches the whole plan.  (A write through `x.data` does not move x's version counter: plan.invalidate() forces
//      the next call to launch.)  Never skipped while the stream is being captured.
struct BatchWatch { PyObject* dict; PyObject* name; PyObject* obj; int64_t version; };
// Consecutive watch entries of ONE dictionary (a quantizer's __dict__).  CPython < 3.12 stamps every dictionary with a version
// tag that changes on ANY modification: while the tag is the one seen at the last full check, every dict[name] is still the
// object it was, so only the tensors' in-place version counters are re-read (54 weights: 324 dictionary lookups saved per
// call).  tag 0 = not valid (a dictionary's tag is never 0 once it has been modified; newer Pythons: always the full check).
struct WatchGroup { size_t first, count; uint64_t tag; };
#if PY_VERSION_HEX < 0x030C0000
inline uint64_t dict_tag(PyObject* d) { return ((PyDictObject*)d)->ma_version_tag; }
#else
inline uint64_t dict_tag(PyObject*) { return 0; }
#endif

struct BatchPlan {
  PyObject_HEAD
  vectorcallfunc vectorcall;
  std::vector<mctq_fq_item>* items;
  std::vector<PyObject*>* refs;          // 4 per item: x, y, scales, zero_points (or Py_None)
  std::vector<std::vector<int64_t>>* sizes;
  std::vector<std::vector<int64_t>>* strides;   // x's strides when the item was packed: same sizes AND strides = same (outer, channels, inner)
  std::vector<int64_t>* axes;            // -1: per tensor
  std::vector<BatchWatch>* watch;
  std::vector<WatchGroup>* groups;
  std::vector<uint8_t>* host_table;
  at::Tensor* dev_table;
  // LUT items (decision-table quantizers): ("lut", x, y, thresholds | None, table, axis | None, eps, thr_div, thr_mul,
  // mult, clip_min, clip_max, step_round[, watch]); y float32, contiguous like x
  std::vector<mctq_lut_item>* lut_items;
  std::vector<PyObject*>* lut_refs;      // 4 per item: x, y, thresholds (or Py_None), table
  std::vector<std::vector<int64_t>>* lut_sizes;
  std::vector<int64_t>* lut_axes;
  std::vector<uint8_t>* lut_host_table;
  at::Tensor* lut_dev_table;
  bool uploaded;
  c10::DeviceIndex device;
  bool versioned;                          // skip the launch when nothing changed since the last one
  bool fresh;                              // the outputs hold the quantization of the inputs as recorded below
  std::vector<uint32_t>* seen;             // per item (affine items first, then LUT items): x version, y version after the launch
  // versioned reuse: the storages the last launch READ stay referenced until the next launch, so that the caching allocator
  // cannot hand their device pointers to a tensor that later becomes a planned weight (`w.data = tmp; w.data = fresh` with no
  // forward in between: same sizes, dtype and version counter; without the reference `fresh` could sit where the launch read)
  std::vector<c10::Storage>* held;
  hipStream_t last_stream;                 // the stream of the last launch: a skip is only valid for work queued behind it
  int64_t launches, skips;
};

// in-place version counter of a tensor; inference tensors have none (they are never skipped over: see `trackable`)
inline uint32_t version_of(const at::Tensor& t) { return t.is_inference() ? 0u : (uint32_t)t._version(); }

inline bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &status) != hipSuccess) { (void)hipGetLastError(); return true; }    // unknown: do not skip
  return status != hipStreamCaptureStatusNone;
}

bool batchplan_upload_lut(BatchPlan* p) {
  const int64_t need = mctq_lutt_batch_pack(p->lut_items->data(), (int32_t)p->lut_items->size(), nullptr, 0);
  if (need < 0) { raise_rc((int)need, "mctq_lutt_batch_pack"); return false; }
  p->lut_host_table->resize((size_t)need);
  const int64_t got = mctq_lutt_batch_pack(p->lut_items->data(), (int32_t)p->lut_items->size(), p->lut_host_table->data(), need);
  if (got != need) { raise_rc((int)got, "mctq_lutt_batch_pack"); return false; }
  if (!p->lut_dev_table->defined() || p->lut_dev_table->numel() < need)
    *p->lut_dev_table = at::Tensor(at::detail::empty_cuda({need}, c10::ScalarType::Byte, c10::Device(c10::kCUDA, p->device), std::nullopt));
  const auto stream = c10::hip::getCurrentHIPStream(p->device);
  c10::hip::memcpy_and_sync(p->lut_dev_table->mutable_data_ptr(), p->lut_host_table->data(), need, hipMemcpyHostToDevice, stream.stream());
  return true;
}

bool batchplan_upload(BatchPlan* p) {
  const int64_t need = mctq_fq_batch_pack(p->items->data(), (int32_t)p->items->size(), nullptr, 0);
  if (need < 0) { raise_rc((int)need, "mctq_fq_batch_pack"); return false; }
  p->host_table->resize((size_t)need);
  const int64_t got = mctq_fq_batch_pack(p->items->data(), (int32_t)p->items->size(), p->host_table->data(), need);
  if (got != need) { raise_rc((int)got, "mctq_fq_batch_pack"); return false; }
  if (!p->dev_table->defined() || p->dev_table->numel() < need)
    *p->dev_table = at::Tensor(at::detail::empty_cuda({need}, c10::ScalarType::Byte, c10::Device(c10::kCUDA, p->device), std::nullopt));
  const auto stream = c10::hip::getCurrentHIPStream(p->device);
  c10::hip::memcpy_and_sync(p->dev_table->mutable_data_ptr(), p->host_table->data(), need, hipMemcpyHostToDevice, stream.stream());
  return true;
}

PyObject* batchplan_vectorcall(PyObject* self, PyObject* const*, size_t nargsf, PyObject* kwnames) {
  HANDLE_TH_ERRORS
  BatchPlan* p = (BatchPlan*)self;
  if (PyVectorcall_NARGS(nargsf) != 0 || (kwnames && PyTuple_GET_SIZE(kwnames))) {
    PyErr_SetString(PyExc_TypeError, "BatchPlan.__call__()");
    return nullptr;
  }
  if (torch::jit::tracer::isTracing()) return not_implemented();
  for (WatchGroup& g : *p->groups) {
    const BatchWatch* w0 = p->watch->data() + g.first;
    const uint64_t tag = dict_tag(w0->dict);
    const bool same_dict = g.tag != 0 && tag == g.tag;               // untouched since the last full check: identities hold
    for (size_t k = 0; k < g.count; ++k) {
      const BatchWatch& w = w0[k];
      PyObject* cur = w.obj;
      if (!same_dict) {
        cur = PyDict_GetItemWithError(w.dict, w.name);                // borrowed
        if (!cur) { if (PyErr_Occurred()) return nullptr; return not_implemented(); }
        if (cur != w.obj) return not_implemented();
      }
      if (w.version >= 0 && (!THPVariable_Check(cur) || (int64_t)THPVariable_Unpack(cur)._version() != w.version))
        return not_implemented();
    }
    g.tag = tag;
  }
  const size_t n = p->items->size();
  bool dirty = !p->uploaded;
  bool changed = !p->versioned || !p->fresh;        // versioned reuse: did anything the last launch read or wrote move on?
  uint32_t* seen = p->seen->data();
  for (size_t i = 0; i < n; ++i) {
    const at::Tensor& x = THPVariable_Unpack((*p->refs)[4 * i]);
    const at::Tensor& y = THPVariable_Unpack((*p->refs)[4 * i + 1]);
    const at::Tensor& sc = THPVariable_Unpack((*p->refs)[4 * i + 2]);
    mctq_fq_item& d = (*p->items)[i];
    if (!changed && (x.is_inference() || version_of(x) != seen[2 * i] || version_of(y) != seen[2 * i + 1])) changed = true;
    const std::vector<int64_t>& sz = (*p->sizes)[i];
    if (x.sizes() != c10::IntArrayRef(sz) || y.sizes() != x.sizes() || !x.is_cuda() || !y.is_cuda() ||
        x.device().index() != p->device || y.device().index() != p->device ||
        !x.unsafeGetTensorImpl()->is_non_overlapping_and_dense() || x.strides() != y.strides() ||
        dtype_code(x.scalar_type()) != d.dtype || y.scalar_type() != x.scalar_type())
      return not_implemented();
    const int64_t axis = (*p->axes)[i];
    if (x.strides() != c10::IntArrayRef((*p->strides)[i])) {         // same sizes, other strides: is the channel view the same?
      int64_t outer, c, inner;
      if (axis < 0) { outer = 1; c = 1; inner = x.numel(); }
      else channel_view(x, axis, &outer, &c, &inner);
      if (outer != d.outer || c != d.channels || inner != d.inner) return not_implemented();
    }
    const int64_t want = axis < 0 ? 1 : d.channels;
    if (sc.scalar_type() != c10::ScalarType::Float || !sc.is_contiguous() || sc.device() != x.device() || sc.numel() != want)
      return not_implemented();
    const void* zp = nullptr;
    if ((*p->refs)[4 * i + 3] != Py_None) {
      const at::Tensor& z = THPVariable_Unpack((*p->refs)[4 * i + 3]);
      if (z.scalar_type() != c10::ScalarType::Int || !z.is_contiguous() || z.device() != x.device() || z.numel() != want)
        return not_implemented();
      zp = z.const_data_ptr();
    }
    const void* xp = x.const_data_ptr();
    void* yp = y.mutable_data_ptr();
    const void* sp = sc.const_data_ptr();
    if (xp != d.x || yp != d.y || sp != (const void*)d.scales || zp != (const void*)d.zero_points) {
      d.x = xp; d.y = yp; d.scales = (const float*)sp; d.zero_points = (const int32_t*)zp;
      dirty = true;
    }
  }
  const size_t nl = p->lut_items->size();
  bool lut_dirty = !p->uploaded;
  for (size_t i = 0; i < nl; ++i) {
    const at::Tensor& x = THPVariable_Unpack((*p->lut_refs)[4 * i]);
    const at::Tensor& y = THPVariable_Unpack((*p->lut_refs)[4 * i + 1]);
    const at::Tensor& tb = THPVariable_Unpack((*p->lut_refs)[4 * i + 3]);
    mctq_lut_item& d = (*p->lut_items)[i];
    if (!changed && (x.is_inference() || version_of(x) != seen[2 * (n + i)] || version_of(y) != seen[2 * (n + i) + 1])) changed = true;
    if (x.sizes() != c10::IntArrayRef((*p->lut_sizes)[i]) || y.sizes() != x.sizes() || !x.is_cuda() || !y.is_cuda() ||
        x.device().index() != p->device || y.device().index() != p->device || !x.is_contiguous() || !y.is_contiguous() ||
        dtype_code(x.scalar_type()) != d.dtype || y.scalar_type() != c10::ScalarType::Float ||
        tb.scalar_type() != c10::ScalarType::Float || !tb.is_contiguous() || tb.device() != x.device() ||
        tb.numel() != 2 * ((int64_t)d.entries + 1))
      return not_implemented();
    const void* tp = nullptr;
    if ((*p->lut_refs)[4 * i + 2] != Py_None) {
      const at::Tensor& t = THPVariable_Unpack((*p->lut_refs)[4 * i + 2]);
      if (t.scalar_type() != c10::ScalarType::Float || !t.is_contiguous() || t.device() != x.device() || t.numel() != d.channels)
        return not_implemented();
      tp = t.const_data_ptr();
    }
    const void* xp = x.const_data_ptr();
    void* yp = y.mutable_data_ptr();
    const void* bp = tb.const_data_ptr();
    if (xp != d.x || yp != (void*)d.y || tp != (const void*)d.thresholds || bp != (const void*)d.table) {
      d.x = xp; d.y = (float*)yp; d.thresholds = (const float*)tp; d.table = (const float*)bp;
      lut_dirty = true;
    }
  }
  if (n > 0 || nl > 0) {
    DeviceScope scope(p->device);
    hipStream_t st = c10::hip::getCurrentHIPStream(p->device).stream();
    // (a call on ANOTHER stream is not ordered behind the launch that filled the outputs: it launches again, on its stream)
    if (p->versioned && !changed && !dirty && !lut_dirty && st == p->last_stream && !stream_is_capturing(st)) {
      ++p->skips;                                    // the outputs ARE the quantization of these inputs: nothing to launch
      Py_RETURN_NONE;
    }
    void* stream = (void*)st;
    p->fresh = false;
    if (n > 0) {
      if (dirty && !batchplan_upload(p)) return nullptr;
      const int rc = mctq_fq_batch_run(p->host_table->data(), p->dev_table->const_data_ptr(), stream);
      if (rc) return raise_rc(rc, "mctq_fq_batch_run");
    }
    if (nl > 0) {
      if (lut_dirty && !batchplan_upload_lut(p)) return nullptr;
      const int rc = mctq_lutt_batch_run(p->lut_host_table->data(), p->lut_dev_table->const_data_ptr(), stream);
      if (rc) return raise_rc(rc, "mctq_lutt_batch_run");
    }
    p->uploaded = true;
    p->last_stream = st;
    ++p->launches;
    // the kernels rewrote every output in place: say so to autograd, and remember what this launch read and wrote
    if (p->versioned) p->held->clear();
    for (size_t i = 0; i < n + nl; ++i) {
      PyObject* const* r = i < n ? &(*p->refs)[4 * i] : &(*p->lut_refs)[4 * (i - n)];
      const at::Tensor& x = THPVariable_Unpack(r[0]);
      const at::Tensor& y = THPVariable_Unpack(r[1]);
      if (!y.is_inference()) y.unsafeGetTensorImpl()->bump_version();
      seen[2 * i] = version_of(x);
      seen[2 * i + 1] = version_of(y);
      if (p->versioned) p->held->push_back(x.storage());
    }
    p->fresh = true;
  }
  Py_RETURN_NONE;
  END_HANDLE_TH_ERRORS
}

void batchplan_dealloc(PyObject* self) {
  BatchPlan* p = (BatchPlan*)self;
  if (p->refs) for (PyObject* o : *p->refs) Py_XDECREF(o);
  if (p->watch) for (BatchWatch& w : *p->watch) { Py_XDECREF(w.dict); Py_XDECREF(w.name); Py_XDECREF(w.obj); }
  if (p->lut_refs) for (PyObject* o : *p->lut_refs) Py_XDECREF(o);
  delete p->items; delete p->refs; delete p->sizes; delete p->strides; delete p->axes; delete p->watch; delete p->groups; delete p->host_table; delete p->dev_table;
  delete p->lut_items; delete p->lut_refs; delete p->lut_sizes; delete p->lut_axes; delete p->lut_host_table; delete p->lut_dev_table;
  delete p->seen;
  delete p->held;
  Py_TYPE(self)->tp_free(self);
}

// plan.invalidate(): the next call launches whatever the version counters say (after a write the counters cannot see)
PyObject* batchplan_invalidate(PyObject* self, PyObject*) {
  ((BatchPlan*)self)->fresh = false;
  Py_RETURN_NONE;
}
// plan.stats() -> (launches, skipped calls)
PyObject* batchplan_stats(PyObject* self, PyObject*) {
  BatchPlan* p = (BatchPlan*)self;
  return Py_BuildValue("(LL)", (long long)p->launches, (long long)p->skips);
}
PyMethodDef batchplan_methods[] = {
    {"invalidate", batchplan_invalidate, METH_NOARGS, nullptr},
    {"stats", batchplan_stats, METH_NOARGS, nullptr},
    {nullptr, nullptr, 0, nullptr}};

// watch = (dict, ((name, object, version), ...)); returns an error text or nullptr
const char* batchplan_add_watch(BatchPlan* p, PyObject* w) {
  if (w == Py_None) return nullptr;
  if (!PyTuple_Check(w) || PyTuple_GET_SIZE(w) != 2 || !PyDict_Check(PyTuple_GET_ITEM(w, 0)) || !PyTuple_Check(PyTuple_GET_ITEM(w, 1)))
    return "watch: (dict, ((name, object, version), ...))";
  PyObject* dict = PyTuple_GET_ITEM(w, 0);
  PyObject* ents = PyTuple_GET_ITEM(w, 1);
  for (Py_ssize_t j = 0; j < PyTuple_GET_SIZE(ents); ++j) {
    PyObject* e = PyTuple_GET_ITEM(ents, j);
    if (!PyTuple_Check(e) || PyTuple_GET_SIZE(e) != 3 || !PyUnicode_Check(PyTuple_GET_ITEM(e, 0)) || !PyLong_Check(PyTuple_GET_ITEM(e, 2)))
      return "watch entry: (name, object, version)";
    BatchWatch bw;
    bw.dict = dict; bw.name = PyTuple_GET_ITEM(e, 0); bw.obj = PyTuple_GET_ITEM(e, 1);
    bw.version = PyLong_AsLongLong(PyTuple_GET_ITEM(e, 2));
    Py_INCREF(bw.dict); Py_INCREF(bw.name); Py_INCREF(bw.obj);
    p->watch->push_back(bw);
  }
  return nullptr;
}

PyObject* batchplan_new(PyTypeObject* type, PyObject* args, PyObject*) {
  HANDLE_TH_ERRORS
  if (PyTuple_GET_SIZE(args) != 1 && PyTuple_GET_SIZE(args) != 2) { PyErr_SetString(PyExc_TypeError, "BatchPlan(items[, versioned])"); return nullptr; }
  int versioned = 0;
  if (PyTuple_GET_SIZE(args) == 2 && (versioned = PyObject_IsTrue(PyTuple_GET_ITEM(args, 1))) < 0) return nullptr;
  PyObject* seq = PySequence_Fast(PyTuple_GET_ITEM(args, 0), "BatchPlan expects a sequence of tuples");
  if (!seq) return nullptr;
  BatchPlan* p = (BatchPlan*)type->tp_alloc(type, 0);
  if (!p) { Py_DECREF(seq); return nullptr; }
  p->vectorcall = batchplan_vectorcall;
  p->items = new std::vector<mctq_fq_item>(); p->refs = new std::vector<PyObject*>();
  p->sizes = new std::vector<std::vector<int64_t>>(); p->strides = new std::vector<std::vector<int64_t>>();
  p->axes = new std::vector<int64_t>();
  p->watch = new std::vector<BatchWatch>(); p->groups = new std::vector<WatchGroup>();
  p->host_table = new std::vector<uint8_t>(); p->dev_table = new at::Tensor();
  p->lut_items = new std::vector<mctq_lut_item>(); p->lut_refs = new std::vector<PyObject*>();
  p->lut_sizes = new std::vector<std::vector<int64_t>>(); p->lut_axes = new std::vector<int64_t>();
  p->lut_host_table = new std::vector<uint8_t>(); p->lut_dev_table = new at::Tensor();
  p->uploaded = false;
  p->device = -1;
  p->versioned = versioned != 0; p->fresh = false; p->launches = 0; p->skips = 0;
  p->seen = new std::vector<uint32_t>();
  p->held = new std::vector<c10::Storage>();
  p->last_stream = nullptr;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  const char* err = nullptr;
  for (Py_ssize_t i = 0; i < n && !err; ++i) {
    PyObject* it = PySequence_Fast_GET_ITEM(seq, i);
    if (PyTuple_Check(it) && PyTuple_GET_SIZE(it) >= 13 && PyUnicode_Check(PyTuple_GET_ITEM(it, 0))) {
      // ---- a LUT item ----
      if (PyTuple_GET_SIZE(it) > 14 || PyUnicode_CompareWithASCIIString(PyTuple_GET_ITEM(it, 0), "lut") != 0) {
        err = "LUT item: ('lut', x, y, thresholds, table, axis, eps, thr_div, thr_mul, mult, clip_min, clip_max, step_round[, watch])"; break;
      }
      int dt, dty;
      const at::Tensor* xp = eligible(PyTuple_GET_ITEM(it, 1), &dt);
      const at::Tensor* yp = eligible(PyTuple_GET_ITEM(it, 2), &dty);
      if (!xp || !yp || dt == MCTQ_DT_F64 || yp->scalar_type() != c10::ScalarType::Float || !xp->is_contiguous() ||
          !yp->is_contiguous() || xp->sizes() != yp->sizes() || xp->device() != yp->device()) {
        err = "LUT item: x a contiguous float32 / float16 / bfloat16 HIP tensor, y float32 of the same shape"; break;
      }
      double v[6];
      bool okv = true;
      for (int k = 0; k < 6 && okv; ++k) okv = as_double(PyTuple_GET_ITEM(it, 6 + k), &v[k]);
      int64_t sr, axis = -1;
      PyObject* axis_o = PyTuple_GET_ITEM(it, 5);
      if (!okv || !as_i64(PyTuple_GET_ITEM(it, 12), &sr) || (axis_o != Py_None && !as_i64(axis_o, &axis))) {
        Py_DECREF(seq); Py_DECREF(p); return nullptr;
      }
      if (axis_o != Py_None && (axis < 0 || axis >= xp->dim())) { err = "axis out of range"; break; }
      if (p->device < 0) p->device = xp->device().index();
      if (xp->device().index() != p->device) { err = "all tensors of one plan must be on the same device"; break; }
      const at::Tensor* tb = param_tensor(PyTuple_GET_ITEM(it, 4), *xp, c10::ScalarType::Float, -1);
      if (!tb || tb->numel() < 4 || (tb->numel() & 1)) { err = "LUT item: table must be the float32 decision table on x's device"; break; }
      const at::Tensor* th = nullptr;
      mctq_lut_item d;
      d.x = xp->const_data_ptr(); d.y = yp->mutable_data_ptr<float>();
      if (axis_o == Py_None) { d.outer = 1; d.channels = 1; d.inner = xp->numel(); }
      else channel_view(*xp, axis, &d.outer, &d.channels, &d.inner);
      if (PyTuple_GET_ITEM(it, 3) != Py_None) {
        th = param_tensor(PyTuple_GET_ITEM(it, 3), *xp, c10::ScalarType::Float, axis_o == Py_None ? 1 : xp->size(axis));
        if (!th) { err = "LUT item: thresholds must be a contiguous float32 tensor on x's device with one entry per channel"; break; }
      } else if (axis_o != Py_None) { err = "LUT item: a per-channel item needs thresholds"; break; }
      d.thresholds = th ? th->const_data_ptr<float>() : nullptr;
      d.table = tb->const_data_ptr<float>();
      d.entries = (int32_t)(tb->numel() / 2 - 1);
      d.eps = (float)v[0]; d.thr_div = (float)v[1]; d.thr_mul = (float)v[2]; d.mult = (float)v[3]; d.clip_min = (float)v[4]; d.clip_max = (float)v[5];
      d.dtype = dt; d.step_round = (int32_t)sr;
      p->lut_items->push_back(d);
      p->lut_sizes->push_back(xp->sizes().vec());
      p->lut_axes->push_back(axis_o == Py_None ? -1 : axis);
      for (int k = 1; k <= 4; ++k) { PyObject* o = PyTuple_GET_ITEM(it, k); Py_INCREF(o); p->lut_refs->push_back(o); }
      if (PyTuple_GET_SIZE(it) == 14) err = batchplan_add_watch(p, PyTuple_GET_ITEM(it, 13));
      continue;
    }
    if (!PyTuple_Check(it) || (PyTuple_GET_SIZE(it) != 7 && PyTuple_GET_SIZE(it) != 8)) {
      err = "item: (x, y, scales, zero_points, axis, quant_min, quant_max[, watch])"; break;
    }
    int dt, dty;
    const at::Tensor* xp = eligible(PyTuple_GET_ITEM(it, 0), &dt);
    const at::Tensor* yp = eligible(PyTuple_GET_ITEM(it, 1), &dty);
    int64_t qmin, qmax, axis = -1;
    PyObject* axis_o = PyTuple_GET_ITEM(it, 4);
    if (!as_i64(PyTuple_GET_ITEM(it, 5), &qmin) || !as_i64(PyTuple_GET_ITEM(it, 6), &qmax) || (axis_o != Py_None && !as_i64(axis_o, &axis))) {
      Py_DECREF(seq); Py_DECREF(p); return nullptr;
    }
    if (!xp || !yp) { err = "x and y must be plain dense HIP tensors of a supported dtype"; break; }
    if (dt != dty || xp->sizes() != yp->sizes() || xp->strides() != yp->strides() || xp->device() != yp->device()) {
      err = "y must have x's dtype, shape, strides and device"; break;
    }
    if (axis_o != Py_None && (axis < 0 || axis >= xp->dim())) { err = "axis out of range"; break; }
    if (p->device < 0) p->device = xp->device().index();
    if (xp->device().index() != p->device) { err = "all tensors of one plan must be on the same device"; break; }
    const int64_t want = axis_o == Py_None ? 1 : xp->size(axis);
    const at::Tensor* sp = param_tensor(PyTuple_GET_ITEM(it, 2), *xp, c10::ScalarType::Float, want);
    const at::Tensor* zp = nullptr;
    if (PyTuple_GET_ITEM(it, 3) != Py_None) {
      zp = param_tensor(PyTuple_GET_ITEM(it, 3), *xp, c10::ScalarType::Int, want);
      if (!zp) { err = "zero_points must be a contiguous int32 tensor on x's device with one entry per channel"; break; }
    }
    if (!sp) { err = "scales must be a contiguous float32 tensor on x's device with one entry per channel"; break; }
    if (dt == MCTQ_DT_F64 && axis_o == Py_None && !zp) { err = "a float64 per-tensor item needs zero_points"; break; }
    mctq_fq_item d;
    d.x = xp->const_data_ptr(); d.y = yp->mutable_data_ptr();
    if (axis_o == Py_None) { d.outer = 1; d.channels = 1; d.inner = xp->numel(); }
    else channel_view(*xp, axis, &d.outer, &d.channels, &d.inner);
    d.scales = sp->const_data_ptr<float>();
    d.zero_points = zp ? zp->const_data_ptr<int32_t>() : nullptr;
    d.quant_min = (int32_t)qmin; d.quant_max = (int32_t)qmax; d.dtype = dt;
    d.flags = axis_o == Py_None ? MCTQ_FQ_ITEM_PER_TENSOR : 0;
    p->items->push_back(d);
    p->sizes->push_back(xp->sizes().vec());
    p->strides->push_back(xp->strides().vec());
    p->axes->push_back(axis_o == Py_None ? -1 : axis);
    for (int k = 0; k < 4; ++k) { PyObject* o = PyTuple_GET_ITEM(it, k); Py_INCREF(o); p->refs->push_back(o); }
    if (PyTuple_GET_SIZE(it) == 8) err = batchplan_add_watch(p, PyTuple_GET_ITEM(it, 7));
  }
  Py_DECREF(seq);
  if (err) { Py_DECREF(p); PyErr_Format(PyExc_TypeError, "BatchPlan: %s", err); return nullptr; }
  p->seen->assign(2 * (p->items->size() + p->lut_items->size()), 0u);
  for (size_t i = 0; i < p->watch->size();) {                       // runs of entries that watch the same dictionary
    size_t j = i;
    while (j < p->watch->size() && (*p->watch)[j].dict == (*p->watch)[i].dict) ++j;
    p->groups->push_back(WatchGroup{i, j - i, 0});
    i = j;
  }
  return (PyObject*)p;
  END_HANDLE_TH_ERRORS
}

PyTypeObject BatchPlanType = {PyVarObject_HEAD_INIT(nullptr, 0)};

PyObject* py_abi_version(PyObject*, PyObject*) { return PyLong_FromLong(mctq_abi_version()); }

#ifndef MCTQ_BINDING_ID
#define MCTQ_BINDING_ID "unstamped"
#endif
// "MCTQ_BINDING_ID=<id>": content hash of what this module was built from (hip/build.py: binding_build_id), also found
// by scanning the file
PyObject* py_build_id(PyObject*, PyObject*) {
  static const char text[] = "MCTQ_BINDING_ID=" MCTQ_BINDING_ID;
  return PyUnicode_FromString(text + sizeof("MCTQ_BINDING_ID=") - 1);
}

PyMethodDef methods[] = {
    {"fq_per_tensor", (PyCFunction)(void (*)(void))py_fq_per_tensor, METH_FASTCALL, nullptr},
    {"fq_per_channel", (PyCFunction)(void (*)(void))py_fq_per_channel, METH_FASTCALL, nullptr},
    {"fq_per_tensor_tqp", (PyCFunction)(void (*)(void))py_fq_per_tensor_tqp, METH_FASTCALL, nullptr},
    {"lutt_per_tensor", (PyCFunction)(void (*)(void))py_lutt_per_tensor, METH_FASTCALL, nullptr},
    {"lutt_per_channel", (PyCFunction)(void (*)(void))py_lutt_per_channel, METH_FASTCALL, nullptr},
    {"fq_batched", (PyCFunction)(void (*)(void))py_fq_batched, METH_FASTCALL, nullptr},
    {"abi_version", py_abi_version, METH_NOARGS, nullptr},
    {"build_id", py_build_id, METH_NOARGS, nullptr},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_mctq_torch",
                         "compiled binding of libmctq_hip.so's hot entry points (see mctq_torch.cpp)", -1, methods};

}  // namespace

PyMODINIT_FUNC PyInit__mctq_torch(void) {
  AffinePlanType.tp_name = "_mctq_torch.AffinePlan";
  AffinePlanType.tp_basicsize = sizeof(AffinePlan);
  AffinePlanType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL;
  AffinePlanType.tp_new = plan_new;
  AffinePlanType.tp_dealloc = plan_dealloc;
  AffinePlanType.tp_call = PyVectorcall_Call;
  AffinePlanType.tp_vectorcall_offset = offsetof(AffinePlan, vectorcall);
  if (PyType_Ready(&AffinePlanType) < 0) return nullptr;
  LutPlanType.tp_name = "_mctq_torch.LutPlan";
  LutPlanType.tp_basicsize = sizeof(LutPlan);
  LutPlanType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL;
  LutPlanType.tp_new = lutplan_new;
  LutPlanType.tp_dealloc = lutplan_dealloc;
  LutPlanType.tp_call = PyVectorcall_Call;
  LutPlanType.tp_vectorcall_offset = offsetof(LutPlan, vectorcall);
  if (PyType_Ready(&LutPlanType) < 0) return nullptr;
  HolderCallType.tp_name = "_mctq_torch.HolderCall";
  HolderCallType.tp_basicsize = sizeof(HolderCall);
  HolderCallType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL | Py_TPFLAGS_HAVE_GC;
  HolderCallType.tp_new = holdercall_new;
  HolderCallType.tp_dealloc = holdercall_dealloc;
  HolderCallType.tp_traverse = holdercall_traverse;
  HolderCallType.tp_clear = holdercall_clear;
  HolderCallType.tp_call = PyVectorcall_Call;
  HolderCallType.tp_vectorcall_offset = offsetof(HolderCall, vectorcall);
  if (PyType_Ready(&HolderCallType) < 0) return nullptr;
  kw_compiled = PyUnicode_InternFromString("_compiled_call_impl");
  kw_quantizer = PyUnicode_InternFromString("activation_holder_quantizer");
  kw_bypass = PyUnicode_InternFromString("quantization_bypass");
  kw_plan = PyUnicode_InternFromString("_plan");
  BatchPlanType.tp_name = "_mctq_torch.BatchPlan";
  BatchPlanType.tp_basicsize = sizeof(BatchPlan);
  BatchPlanType.tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL;
  BatchPlanType.tp_new = batchplan_new;
  BatchPlanType.tp_methods = batchplan_methods;
  BatchPlanType.tp_dealloc = batchplan_dealloc;
  BatchPlanType.tp_call = PyVectorcall_Call;
  BatchPlanType.tp_vectorcall_offset = offsetof(BatchPlan, vectorcall);
  if (PyType_Ready(&BatchPlanType) < 0) return nullptr;
  PyObject* m = PyModule_Create(&moduledef);
  if (!m) return nullptr;
  Py_INCREF(&AffinePlanType);
  if (PyModule_AddObject(m, "AffinePlan", (PyObject*)&AffinePlanType) < 0) { Py_DECREF(m); return nullptr; }
  Py_INCREF(&LutPlanType);
  if (PyModule_AddObject(m, "LutPlan", (PyObject*)&LutPlanType) < 0) { Py_DECREF(m); return nullptr; }
  Py_INCREF(&BatchPlanType);
  if (PyModule_AddObject(m, "BatchPlan", (PyObject*)&BatchPlanType) < 0) { Py_DECREF(m); return nullptr; }
  Py_INCREF(&HolderCallType);
  if (PyModule_AddObject(m, "HolderCall", (PyObject*)&HolderCallType) < 0) { Py_DECREF(m); return nullptr; }
  return m;
}
